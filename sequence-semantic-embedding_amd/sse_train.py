"""`sse_train` command (reference `sse_train.py:60-248`): same flags, log lines,
checkpoint names (SSE-LSTM.ckpt-BestEver / -final / -epoch-N), learning-rate
decay and early-stop rules, per-epoch re-index + evaluation; the model runs on
the MI355X through the C ABI."""
import logging
import os
import sys
import time
from logging import handlers

from . import flags, sse_data, sse_index
from .sse_evaluator import Evaluator
from .sse_model import Session, SSEModel, get_checkpoint_state

FLAGS = flags.FlagSet("sse_train", [
    ("learning_rate", float, 0.9, "Learning rate."),
    ("learning_rate_decay_factor", float, 0.99, "Learning rate decays by this much."),
    ("batch_size", int, 64, "Batch size to use during training(positive pair count based)."),
    ("embedding_size", int, 50, "Size of word embedding vector."),
    ("encoding_size", int, 64, "Size of sequence encoding vector."),
    ("src_cell_size", int, 96, "LSTM cell size in source RNN model."),
    ("tgt_cell_size", int, 96, "LSTM cell size in target RNN model."),
    ("num_layers", int, 1, "Number of layers in the model (unused, as in the reference)."),
    ("vocab_size", int, 32000, "Target size when a vocabulary has to be built."),
    ("max_seq_length", int, 80, "max number of words in each source or target sequence."),
    ("max_epoc", int, 30, "max epoc number for training procedure."),
    ("predict_nbest", int, 10, "max top N for evaluation prediction."),
    ("task_type", str, "classification", "classification, ranking, qna, crosslingual"),
    ("data_dir", str, "rawdata-classification", "Data directory"),
    ("model_dir", str, "models-classification", "Trained model directory."),
    ("rawfilename", str, "targetIDs", "raw target sequence file to be indexed"),
    ("encodedIndexFile", str, "targetEncodingIndex.tsv", "target sequece encoding index file."),
    ("device", str, "0", "GPU ordinal."),
    ("network_mode", str, "dual-encoder", "source-encoder-only, dual-encoder, shared-encoder, source_only_cnn"),
    ("steps_per_checkpoint", int, 200, "How many training steps to do per checkpoint."),
    ("seed", int, -1, "seed for batch sampling and initialisation (-1: unseeded, like the reference)"),
    ("max_steps", int, 0, "stop after this many steps (0: no limit; for smoke runs)"),
    ("device_corpus", int, 1, "1: the padded corpora are uploaded once and a step ships row numbers; 0: token-id feed dicts"),
    ("cnn_bf16", int, 0, "source_only_cnn: 1 = mixed precision (convolution on bf16-rounded embeddings / filters, float32 masters)"),
    ("train_x3", int, 0, "LSTM modes: 0 (default) = float32 MFMA throughout, the reference's arithmetic; 1 = opt in to the forward / "
                         "BPTT / weight-gradient GEMMs of the train step on the bf16 matrix pipe with hi + lo split float32 "
                         "operands (~4e-6 relative per product, ~2x faster)"),
])


def create_model(f, session, targetSpaceSize, vocabsize, forward_only):
    """sse_train.py:96-121."""
    params = {"max_seq_length": f.max_seq_length, "vocab_size": vocabsize, "embedding_size": f.embedding_size,
              "encoding_size": f.encoding_size, "learning_rate": f.learning_rate,
              "learning_rate_decay_factor": f.learning_rate_decay_factor, "src_cell_size": f.src_cell_size,
              "tgt_cell_size": f.tgt_cell_size, "network_mode": f.network_mode, "predict_nbest": f.predict_nbest,
              "targetSpaceSize": targetSpaceSize, "forward_only": forward_only}
    sse_data.save_model_configs(f.model_dir, params)
    model = SSEModel(params, device=int(f.device))
    if getattr(f, "cnn_bf16", 0):
        model.handle.set_option("cnn_bf16", 1)                     # fails loudly outside source_only_cnn
    if getattr(f, "train_x3", 0) and f.network_mode != "source_only_cnn":   # opt-in split-operand train step (library default: float32)
        for opt in ("train_dk_x3", "train_fwd_x3", "train_bwd_x3"):
            model.handle.set_option(opt, 1)
    ckpt = get_checkpoint_state(f.model_dir)
    if ckpt:
        logging.info("Reading model parameters from %s" % ckpt)
        model.saver.restore(session, ckpt)
    else:
        if forward_only:
            raise FileNotFoundError("Error!!!Could not load any model from specified folder: %s" % f.model_dir)
        logging.info("Created model with fresh parameters.")
        model.init_variables(seed=None if f.seed < 0 else f.seed)
    return model


def set_up_logging(model_dir):
    os.makedirs(model_dir, exist_ok=True)
    log = logging.getLogger("")
    log.setLevel(logging.DEBUG)
    fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s", datefmt="%m/%d/%Y %I:%M:%S %p")
    ch = logging.StreamHandler(sys.stdout)
    ch.setFormatter(fmt)
    log.addHandler(ch)
    fh = handlers.RotatingFileHandler(os.path.join(model_dir, "TrainingLog.txt"), maxBytes=1048576 * 20, backupCount=7)
    fh.setFormatter(fmt)
    log.addHandler(fh)


def train(f):
    logging.info("Preparing Train & Eval data in %s" % f.data_dir)
    for d in (f.data_dir, f.model_dir):
        os.makedirs(d, exist_ok=True)
    data = sse_data.Data(f.model_dir, f.data_dir, f.vocab_size, f.max_seq_length,
                         seed=None if f.seed < 0 else f.seed, log=logging.info)
    epoc_steps = len(data.rawTrainPosCorpus) / f.batch_size
    logging.info("Training Data: %d total positive samples, each epoch need %d steps"
                 % (len(data.rawTrainPosCorpus), epoc_steps))
    model = create_model(f, None, data.rawnegSetLen, data.vocab_size, False)
    sess = Session(model)
    summary_op = model.add_summaries()
    table_tgt = f.network_mode in ("source_only_cnn", "source-encoder-only")
    if f.device_corpus:
        src_corpus, tgt_corpus = data.corpus_matrices()
        model.handle.corpus_upload(0, src_corpus)
        if not table_tgt:
            model.handle.corpus_upload(1, tgt_corpus)
    step_time, loss, train_acc = 0.0, 0.0, 0.0
    current_step, previous_accuracies, stop = 0, [], False
    checkpoint_path = os.path.join(f.model_dir, "SSE-LSTM.ckpt")
    for epoch in range(f.max_epoc):
        epoc_start = time.time()
        for _ in range(int(epoc_steps)):
            start = time.time()
            model.set_forward_only(False)
            if f.device_corpus:
                # the padded corpora live on the device: a step ships 2*batch row numbers, not token-id matrices
                src_rows, tgt_rows, labels = data.get_train_batch_rows(f.batch_size, vectorized=f.seed < 0)
                step_loss, step_train_acc = model.handle.train_step_rows(src_rows, tgt_rows, labels)
            else:
                src, tgt, labels = data.get_train_batch_arrays(f.batch_size, target_rows=table_tgt)
                d = model.get_train_feed_dict(src, tgt, labels)
                _, _, step_loss, step_train_acc = sess.run([model.train, summary_op, model.loss, model.train_acc], feed_dict=d)
            step_time += (time.time() - start) / f.steps_per_checkpoint
            loss += step_loss / f.steps_per_checkpoint
            train_acc += step_train_acc / f.steps_per_checkpoint
            current_step += 1
            if f.max_steps and current_step >= f.max_steps:
                stop = True
            if current_step % f.steps_per_checkpoint == 0:
                logging.info("global epoc: %.3f, global step %d, learning rate %.4f step-time:%.2f loss:%.4f train_binary_acc:%.4f "
                             % (float(model.global_step.eval()) / float(epoc_steps), model.global_step.eval(),
                                model.learning_rate.eval(), step_time, step_loss, train_acc))
                # sse_train.py:199-214: decay after no improvement over the last 5 windows, keep the best, early stop
                if len(previous_accuracies) > 6 and train_acc < min(previous_accuracies[-5:]):
                    sess.run(model.learning_rate_decay_op)
                previous_accuracies.append(train_acc)
                if train_acc == max(previous_accuracies):
                    logging.info("Better Accuracy %.4f found. Saving current best model ..." % train_acc)
                    model.save(sess, checkpoint_path + "-BestEver")
                else:
                    logging.info("Best Accuracy is: %.4f, while current round is: %.4f" % (max(previous_accuracies), train_acc))
                    logging.info("skip saving model ...")
                if epoch > 10 and train_acc < min(previous_accuracies[-5:]):
                    p = model.save(sess, checkpoint_path + "-final")
                    logging.info("After around %d Epocs no further improvement, Training finished, wrote checkpoint to %s." % (epoch, p))
                    break
                step_time, loss, train_acc = 0.0, 0.0, 0.0
            if stop:
                break
        logging.info("\n\n\nepoch# %d  took %f hours" % (epoch, (time.time() - epoc_start) / 3600.0))
        if (f.task_type not in ["ranking", "crosslingual"]) or ((epoch + 1) % 20 == 0) or stop:
            model.set_forward_only(True)
            idx_file = os.path.join(f.model_dir, f.encodedIndexFile)
            sse_index.createIndexFile(model, data.encoder, os.path.join(f.model_dir, f.rawfilename), f.max_seq_length,
                                      idx_file, sess, batchsize=1000, row_of=data.target_row)
            acc1, acc3, acc10 = Evaluator(model, data.rawEvalCorpus, idx_file, sess).eval()
            logging.info("epoc#%d, task specific evaluation: top 1/3/10 accuracies: %f / %f / %f \n\n\n" % (epoch, acc1, acc3, acc10))
        model.save(sess, checkpoint_path + "-epoch-%d" % epoch)
        if previous_accuracies:
            logging.info("So far best ever model training binary accuracy is: %.4f " % max(previous_accuracies))
        if stop:
            break


def main(argv=None):
    f = FLAGS.parse(sys.argv[1:] if argv is None else argv)
    if not f.data_dir or not f.model_dir:
        raise SystemExit("--data_dir and --model_dir must be specified.")
    set_up_logging(f.model_dir)
    train(f)


if __name__ == "__main__":
    main()
