"""`sse_index` command: encode every target with the target encoder and write
targetEncodingIndex.tsv (reference `sse_index.py:55-126`; same flags, same file
format `id \\t original-case sentence \\t str(np.float32),...`)."""
import codecs
import math
import os
import sys

import numpy as np

from . import flags, index_io, sse_data, sse_text
from .sse_model import Session, SSEModel, get_checkpoint_state

FLAGS = flags.FlagSet("sse_index", [
    ("idx_model_dir", str, "models-classification", "Trained model directory."),
    ("idx_rawfilename", str, "targetIDs", "raw target sequence file to be indexed"),
    ("idx_encodedIndexFile", str, "targetEncodingIndex.tsv", "target sequece encoding index file."),
    ("device", str, "0", "GPU ordinal"),
])


def createIndexFile(model, encoder, rawfile, max_seq_len, encodeIndexFile, session, batchsize=10000, row_of=None):
    if not os.path.exists(rawfile):
        raise FileNotFoundError("Error!! Could not find raw target file to be indexed!! :%s" % rawfile)
    lines = codecs.open(rawfile, "r", "utf-8").readlines()
    cnt = 0
    print("Start indexing whole target space entries with current model ...")
    if model.network_mode in ("source_only_cnn", "source-encoder-only"):
        # no target sequence encoder: norm_tgt_seq_embedding IS the [N,S] variable (sse_model.py:214,233,283);
        # a target id owns the row of its FIRST occurrence among the well-formed lines -- the order in which
        # training numbers the ids (Data.target_row: de-duplicated dict order of fullSetTargetIds); a duplicate id
        # in the file must not shift every later row by one.  row_of(id) overrides when the caller knows better.
        n_rows = int(model.targetSpaceSize)
        first_row = {}
        table = np.vstack(session.run([model.norm_tgt_seq_embedding],
                                      feed_dict=model.get_target_encoding_feed_dict(np.zeros((n_rows, max_seq_len), np.int32))))
        with codecs.open(encodeIndexFile, "w", "utf-8") as out:
            r = 0
            for line in lines:
                cnt += 1
                info = line.strip().split("\t")
                if len(info) != 2:
                    print("Missing field with error line in raw target file: %s " % line)
                    continue
                if info[1] not in first_row:
                    first_row[info[1]] = r
                    r += 1
                row = row_of(info[1]) if row_of else first_row[info[1]]
                if row >= n_rows:
                    raise ValueError("target file has more entries than the model's target matrix (%d rows)" % n_rows)
                out.write(info[1] + "\t" + info[0] + "\t" + index_io.format_rows(table[row:row + 1])[0] + "\n")
        print("Done of all indexing total count:%d" % cnt)
        return
    with codecs.open(encodeIndexFile, "w", "utf-8") as out:
        for b in range(int(math.ceil(len(lines) / float(batchsize)))):
            ids, tids, sents = [], [], []
            for line in lines[b * batchsize:(b + 1) * batchsize]:
                cnt += 1
                info = line.strip().split("\t")
                if len(info) != 2:                                    # sse_index.py:72-74
                    print("Missing field with error line in raw target file: %s " % line)
                    continue
                ids.append(sse_text.pad_tokens(encoder.encode(info[0].lower()), max_seq_len))
                tids.append(info[1])
                sents.append(info[0])
            if not ids:
                continue
            enc = np.vstack(session.run([model.norm_tgt_seq_embedding],
                                        feed_dict=model.get_target_encoding_feed_dict(ids)))
            vecs = index_io.format_rows(enc)          # == ",".join([str(n) for n in enc[i]]), sse_index.py:95
            for i in range(len(sents)):
                out.write(tids[i] + "\t" + sents[i] + "\t" + vecs[i] + "\n")
    print("Done of all indexing total count:%d" % cnt)


def index(model_dir, rawfile, encodeIndexFile, batchsize=10000, device=0):
    if not os.path.exists(model_dir):
        raise FileNotFoundError("Error! Model folder does not exist!! : %s" % model_dir)
    vocab_file = os.path.join(model_dir, "vocabulary.txt")
    if not os.path.exists(vocab_file):
        raise FileNotFoundError("Error!! Could not find vocabulary file for encoder in folder :%s" % model_dir)
    encoder = sse_text.SubwordVocab(vocab_file)
    print("Loaded  vocab size is: %d" % encoder.vocab_size)
    cfg = sse_data.load_model_configs(model_dir)
    model = SSEModel(cfg, device=device)
    ckpt = get_checkpoint_state(model_dir)
    if not ckpt:
        raise FileNotFoundError("Error!!!Could not load any model from specified folder: %s" % model_dir)
    print("Reading model parameters from %s" % ckpt)
    model.saver.restore(None, ckpt)
    createIndexFile(model, encoder, rawfile, int(cfg["max_seq_length"]), encodeIndexFile, Session(model), batchsize)


def main(argv=None):
    f = FLAGS.parse(sys.argv[1:] if argv is None else argv)
    index(f.idx_model_dir, os.path.join(f.idx_model_dir, f.idx_rawfilename),
          os.path.join(f.idx_model_dir, f.idx_encodedIndexFile), device=int(f.device))


if __name__ == "__main__":
    main()
