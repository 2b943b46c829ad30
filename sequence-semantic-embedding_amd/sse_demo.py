"""`sse_demo` command (reference `sse_demo.py:59-146`): read a query per line from
stdin, encode it with the SOURCE encoder -- un-normalised, as the reference
does at :123 -- score it against the index on the GPU and print the top-N.
The line is lower-cased and tokenised WITH its trailing newline, exactly as
the reference does (`encoder.encode(tf.compat.as_str(sentence).lower())`,
:113): the same checkpoint, index and keystrokes give the same token ids."""
import os
import sys

import numpy as np

from . import flags, sse_data, sse_text
from .sse_evaluator import load_index_file
from .sse_model import SSEModel, get_checkpoint_state

FLAGS = flags.FlagSet("sse_demo", [
    ("device", str, "0", "GPU ordinal."),
    ("model_dir", str, "models-classification", "Trained model directory."),
    ("indexFile", str, "targetEncodingIndex.tsv", "Index file inside model_dir."),
])


def demo(f, nbest, stdin=sys.stdin, out=sys.stdout):
    if not os.path.exists(f.model_dir):
        raise FileNotFoundError("Model folder does not exist!!")
    vocab_file = os.path.join(f.model_dir, "vocabulary.txt")
    if not os.path.exists(vocab_file):
        raise FileNotFoundError("Error!! Could not find vocabulary file for encoder in model folder.")
    encoder = sse_text.SubwordVocab(vocab_file)
    index_path = os.path.join(f.model_dir, f.indexFile)
    if not os.path.exists(index_path):
        raise FileNotFoundError("Index file does not exist!!!")
    targetIDs, names, targetEncodings, _ = load_index_file(index_path)
    cfg = sse_data.load_model_configs(f.model_dir)
    model = SSEModel(cfg, device=int(f.device))
    ckpt = get_checkpoint_state(f.model_dir)
    if not ckpt:
        raise FileNotFoundError("Error!!!Could not load any model from specified folder: %s" % f.model_dir)
    print("Reading model parameters from %s" % ckpt, file=out)
    model.saver.restore(None, ckpt)
    model.handle.index_upload(targetEncodings)
    index_gen = model.handle.index_gen
    max_seq_length = int(cfg["max_seq_length"])
    nbest = min(nbest, len(targetIDs))
    out.write("\n\nPlease type some keywords to get related task results.\nType 'exit' to quit demo.\n > ")
    out.flush()
    sentence = stdin.readline()
    while sentence and sentence.strip().lower() != "exit":
        source_tokens = encoder.encode(sentence.lower())
        if len(source_tokens) > max_seq_length - 2:
            print("Input sentence too long, max allowed is %d. Try to increase limit!!!!" % max_seq_length, file=out)
        tokens = sse_text.pad_tokens(source_tokens, max_seq_length)
        model.set_forward_only(True)
        if model.handle.index_gen != index_gen:
            model.handle.index_upload(targetEncodings)
            index_gen = model.handle.index_gen
        # sess.run([model.src_seq_embedding]) + np.dot + getSortedResults[:nbest] (:121-129), encoding kept on the device
        scores, idx = model.handle.encode_score_topk(0, np.array([tokens], np.int32), False, nbest)
        print("Top %s Prediction results are:\n" % nbest, file=out)
        for r in range(nbest):
            tid = targetIDs[idx[0][r]]
            print("top%d:  %s , %f ,  %s " % (r + 1, tid, scores[0][r], names[idx[0][r]]), file=out)
        print("> ", end="", file=out)
        out.flush()
        sentence = stdin.readline()


def main(argv=None):
    f = FLAGS.parse(sys.argv[1:] if argv is None else argv)
    if not f.model_dir:
        raise SystemExit("--model_dir must be specified.")
    demo(f, int(f.rest[0]) if f.rest else 10)


if __name__ == "__main__":
    main()
