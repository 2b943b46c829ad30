from tensorflow import gfile  # noqa: F401
