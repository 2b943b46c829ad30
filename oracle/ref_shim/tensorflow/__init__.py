"""Minimal stand-in for the `tensorflow` module -- TEST INFRASTRUCTURE ONLY.

The reference's host-side code (tokenizer.py, text_encoder.py, data_utils.py,
data.py) imports tensorflow only for file access, logging and a string helper.
This shim provides exactly those so that oracle/make_golden.py can import the
reference's OWN host code in the build container (TensorFlow itself is not
installable here).  Nothing in the product path imports this.
"""
import glob as _glob
import logging as _logging
import os as _os


class _GFile(object):
    @staticmethod
    def Open(name, mode="r"):
        return open(name, mode)

    GFile = Open

    @staticmethod
    def Glob(pattern):
        return _glob.glob(pattern)

    @staticmethod
    def Exists(path):
        return _os.path.exists(path)


class _Logging(object):
    info = staticmethod(_logging.info)
    warning = staticmethod(_logging.warning)
    error = staticmethod(_logging.error)


class _Compat(object):
    @staticmethod
    def as_str(s):
        return s.decode("utf-8") if isinstance(s, bytes) else str(s)

    as_text = as_str


gfile = _GFile()
logging = _Logging()
compat = _Compat()
