"""Serving shell (reference `webserver.py:50-297`): the four GET routes on the MI355X handle.

    MODEL_TYPE=classification INDEX_FILE=targetEncodingIndex.tsv python -m sse_amd.sse_serving [--port 5000]
    (or any WSGI server:  gunicorn 'sse_amd.sse_serving:create_app()')

What a request does in the reference (`webserver.py:124-161` and the three siblings): tokenise + left-pad the query
(`:135-142`), `sess.run` the SOURCE encoder for ONE row -- `/api/classify` fetches the l2-normalised encoding
(`:146`), `/api/search`, `/api/qna`, `/api/crosslingual` the raw one (`:184,225,268`) --, `np.dot` it with every index
row, fully sort, keep `nbest`.  Here the query is encoded and scored in one library call with the encoding staying on
the device (`sse_encode_score_topk`), against the index uploaded once at start-up; concurrent requests are gathered by
a micro-batcher into one launch of the few-queries kernels (<= 32 queries per sweep: the index is streamed once for
all of them).  JSON keys and defaults (`nbest` 8 / 10 / 5 / 10) are the reference's.  No Flask dependency: a plain
WSGI callable (the reference runs Flask under gunicorn, README.md).
"""
import json
import logging
import os
import queue
import threading
import time
from urllib.parse import parse_qs

import numpy as np

from . import sse_data, sse_text
from .sse_evaluator import load_index_file
from .sse_model import SSEModel, get_checkpoint_state

# route -> (query argument, default nbest, normalised encoding?, response key of the query, key of the result list,
#           (id key, name key, score key))                                            webserver.py:124-286
ROUTES = {
    "/api/classify": ("keywords", 8, True, "ReqeustKeywords", "ClassificationResults",
                      ("targetCategoryId", "targetCategoryName", "confidenceScore")),
    "/api/search": ("query", 10, False, "SearchQuery", "SearchRankingResults",
                    ("ListingId", "ListingTitle", "rankingScore")),
    "/api/qna": ("question", 5, False, "Question", "Answers", ("answerDocId", "answerContent", "confidenceScore")),
    "/api/crosslingual": ("query", 10, False, "CrossLingualQuery", "SearchResults",
                          ("documentId", "documentTitle", "confScore")),
}
BANNER = ("Sequence Semantic Embedding NLP toolkit demo webserver. \n For classification task, send GET request with URL "
          "of  /api/classify?keywords=hello kitty sunglasses \n For search relevance ranking task, send GET request with  "
          "/api/search?query=red nike shoes&?nbest=10 \n For question answering task, send GET request with  "
          "/api/qna?question=how does secure pay work&?nbest=5  \n For cross-lingual search task, send  GET request with "
          "/api/crosslingual?query=nike运动鞋&?nbest=10 \n")     # webserver.py:289-291


class Ranker(object):
    """Model + vocabulary + resident index of one model directory (FlaskApp.__init__, webserver.py:54-119)."""

    def __init__(self, model_dir, index_file="targetEncodingIndex.tsv", device=0):
        if not os.path.exists(model_dir):
            raise FileNotFoundError("Model folder %s does not exist!!" % model_dir)
        index_path = os.path.join(model_dir, index_file)
        if not os.path.exists(index_path):
            raise FileNotFoundError("Index File does not exist!!")
        vocab_file = os.path.join(model_dir, "vocabulary.txt")
        if not os.path.exists(vocab_file):
            raise FileNotFoundError("Error!! Could not find vocabulary file for encoder in model folder.")
        self.encoder = sse_text.SubwordVocab(vocab_file)
        self.targetIDs, self.targetNames, encodings, _ = load_index_file(index_path)
        self.modelConfigs = sse_data.load_model_configs(model_dir)
        self.model = SSEModel(self.modelConfigs, device=device)
        ckpt = get_checkpoint_state(model_dir)
        if not ckpt:
            raise FileNotFoundError("Error!!!Could not load any model from specified folder: %s" % model_dir)
        logging.info("loading model from %s" % ckpt)
        self.model.saver.restore(None, ckpt)
        self.max_seq_length = int(self.modelConfigs["max_seq_length"])
        self.model.handle.index_upload(encodings)          # float64 rows as parsed from text; resident from here on
        self._index_gen = self.model.handle.index_gen
        self._encodings = encodings

    def tokens(self, text):
        """webserver.py:135-142: encode(lower-cased text), left-pad / truncate to max_seq_length."""
        return sse_text.pad_tokens(self.encoder.encode(text.lower()), self.max_seq_length)

    def rank(self, token_rows, nbest, normalize):
        """Top-`nbest` (score, target id, target name) per token row; one encode + one sweep for all rows."""
        h = self.model.handle
        if h.index_gen != self._index_gen:                  # somebody scored another index on this handle
            h.index_upload(self._encodings)
            self._index_gen = h.index_gen
        k = max(1, min(int(nbest), len(self.targetIDs)))
        scores, rows = h.encode_score_topk(0, np.asarray(token_rows, np.int32), normalize, k)
        return [[(float(scores[i, j]), self.targetIDs[rows[i, j]], self.targetNames[rows[i, j]]) for j in range(k)]
                for i in range(len(token_rows))]


class MicroBatcher(object):
    """Gathers concurrent requests into one encode + score call: whatever is queued when the worker wakes up (at most
    `max_batch`, waiting at most `max_wait_s` for company) shares one sweep of the index.  Requests of the two
    encoding kinds (normalised / raw) and different nbest go in separate calls of the same wake-up."""

    def __init__(self, ranker, max_batch=32, max_wait_s=0.002):
        self.ranker, self.max_batch, self.max_wait_s = ranker, max_batch, max_wait_s
        self.q = queue.Queue()
        self.batches = 0
        self.requests = 0
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def submit(self, tokens, nbest, normalize, timeout_s=30.0):
        item = {"tokens": tokens, "nbest": nbest, "normalize": normalize, "done": threading.Event()}
        self.q.put(item)
        # never wait forever: a dead worker thread (or a wedged device) must turn into an error response, not a hung request
        deadline = time.monotonic() + timeout_s
        while not item["done"].wait(0.25):
            if not self._t.is_alive():
                raise RuntimeError("the batching worker thread has died; restart the server")
            if time.monotonic() > deadline:
                raise TimeoutError("no answer from the ranking worker within %.0f s" % timeout_s)
        if "error" in item:
            raise item["error"]
        return item["result"]

    def _run(self):
        while True:
            items = []
            try:
                items = [self.q.get()]
                try:
                    while len(items) < self.max_batch:
                        items.append(self.q.get(timeout=self.max_wait_s))
                except queue.Empty:
                    pass
                groups = {}
                for it in items:
                    groups.setdefault((bool(it["normalize"]), int(it["nbest"])), []).append(it)
                for (norm, nbest), grp in groups.items():
                    try:
                        res = self.ranker.rank([it["tokens"] for it in grp], nbest, norm)
                        for it, r in zip(grp, res):
                            it["result"] = r
                    except Exception as e:                       # noqa: BLE001  (delivered to the waiting request)
                        for it in grp:
                            it["error"] = e
                    self.batches += 1
                    self.requests += len(grp)
                    for it in grp:
                        it["done"].set()
            except Exception as e:                               # noqa: BLE001
                # anything else (a malformed item, ...): every request of this wake-up gets the error, the worker lives on
                for it in items:
                    if isinstance(it, dict) and "done" in it and not it["done"].is_set():
                        it["error"] = e
                        it["done"].set()


def handle_request(path, args, rank_fn, tokens_fn):
    """Route logic shared by the WSGI app and the tests: returns (status, body dict | str).
    `args`: {name: value}; `rank_fn(tokens, nbest, normalize)` -> [(score, id, name)]."""
    if path == "/":
        return 200, BANNER
    route = ROUTES.get(path)
    if route is None:
        return 404, "Not Found"
    arg, default_nbest, normalize, qkey, rkey, (idk, namek, scorek) = route
    text = args.get(arg)
    if text is None:
        return 400, "missing query argument '%s'" % arg
    # (the URLs documented in webserver.py:126,166 say '&?nbest=': that argument is named '?nbest', which the
    # reference does not look at -- it answers with the default count; same here)
    nbest = int(args["nbest"]) if "nbest" in args else default_nbest
    ranked = rank_fn(tokens_fn(text), nbest, normalize)
    results = []
    for i, (score, tid, name) in enumerate(ranked):
        logging.info("top%d:  %s , %f ,  %s " % (i + 1, tid, score, name))
        results.append({idk: tid, namek: name, scorek: float(score)})
    return 200, {qkey: text, rkey: results}


def create_app(model_dir=None, index_file=None, device=0, ranker=None, batch=True):
    """WSGI application.  Environment as the reference: MODEL_TYPE (default 'classification') -> model directory
    'models-<MODEL_TYPE>', INDEX_FILE (default targetEncodingIndex.tsv) -- webserver.py:58-60."""
    if ranker is None:
        model_dir = model_dir or ("models-" + os.environ.get("MODEL_TYPE", "classification"))
        index_file = index_file or os.environ.get("INDEX_FILE", "targetEncodingIndex.tsv")
        ranker = Ranker(model_dir, index_file, device)
    batcher = MicroBatcher(ranker) if batch else None

    def rank_fn(tokens, nbest, normalize):
        if batcher is not None:
            return batcher.submit(tokens, nbest, normalize)
        return ranker.rank([tokens], nbest, normalize)[0]

    def app(environ, start_response):
        args = {k: v[0] for k, v in parse_qs(environ.get("QUERY_STRING", ""), keep_blank_values=True).items()}
        try:
            status, body = handle_request(environ.get("PATH_INFO", "/"), args, rank_fn, ranker.tokens)
        except Exception as e:                               # noqa: BLE001
            logging.exception("request failed")
            status, body = 500, "%s: %s" % (type(e).__name__, e)
        if isinstance(body, dict):
            data, ctype = json.dumps(body).encode("utf-8"), "application/json"
        else:
            data, ctype = body.encode("utf-8"), "text/html; charset=utf-8"
        reason = {200: "OK", 400: "Bad Request", 404: "Not Found", 500: "Internal Server Error"}[status]
        start_response("%d %s" % (status, reason), [("Content-Type", ctype), ("Content-Length", str(len(data)))])
        return [data]

    app.ranker, app.batcher = ranker, batcher
    return app


def main(argv=None):
    import argparse
    from socketserver import ThreadingMixIn
    from wsgiref.simple_server import WSGIServer, make_server

    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=5000)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")

    class Server(ThreadingMixIn, WSGIServer):
        daemon_threads = True

    srv = make_server(a.host, a.port, create_app(device=a.device), server_class=Server)
    logging.info("serving on %s:%d" % (a.host, a.port))
    srv.serve_forever()


if __name__ == "__main__":
    main()
