#!/usr/bin/env python
"""Convert a checkpoint written by the reference (TensorFlow 1.x Saver) into the .npz the MI355X build loads, on a
machine that HAS TensorFlow (any 1.x / 2.x):

    python tools/tf_checkpoint_to_npz.py models-classification

For every checkpoint prefix in the directory it writes <prefix>.npz holding each variable under its TF name
(`word_embedding`, `source_encoder/rnn/basic_lstm_cell/kernel`, ..., `<name>/Adagrad`, `learning_rate`, `global_step`)
-- the names ARE the interface (`sse_set_variable`), so this is a dictionary copy.  Without TensorFlow use
`python -m sse_amd.tf_checkpoint <dir>` (pure-Python reader of the V2 bundle format)."""
import glob
import os
import sys

import numpy as np


def main():
    import tensorflow as tf
    d = sys.argv[1]
    prefixes = sorted(p[:-len(".index")] for p in glob.glob(os.path.join(d, "*.index")))
    for prefix in prefixes:
        reader = tf.train.load_checkpoint(prefix)
        arrays = {}
        for name in reader.get_variable_to_shape_map():
            a = reader.get_tensor(name)
            if name == "global_step":
                arrays[name] = np.int64(a)
            elif name == "learning_rate":
                arrays[name] = np.float32(a)
            elif a.dtype == np.float32:
                arrays[name] = a
        np.savez(prefix + ".npz", **arrays)
        print("%s.npz: %d arrays" % (prefix, len(arrays)))


if __name__ == "__main__":
    main()
