"""Host-side string work of the reference's callers -- importable WITHOUT torch and without the HIP library.

* `split_and_decode(ids, special_token_ids, tokenizer)`: reference modules/transformers/longitudinal_model/modelling_longitudinal.py:413-457
  (`split_and_decode_sections`, quirk Q9) on a numpy id matrix; all sections of the batch are decoded by ONE call into the `tokenizers` library.
* `StringWorker`: the string round trip of an SCST step (scst/gt_prompt.py:90-91,120-128,192-197: ids -> findings / impression strings ->
  f"{findings} {impression}" -> reward tokenizer) in a CHILD PROCESS, so that it runs while the parent's Python thread queues the re-scoring pass on
  the GPU. A thread cannot do that: the `tokenizers` bindings keep the GIL (measured: tokenizer work + a pure-Python loop take the sum of their times
  from two threads), and the parent spends those milliseconds inside Python launching kernels. The child is `python -m cxrmate_amd.strings`
  (started once, numpy + tokenizers only, no torch, no GPU), spoken to over its stdin / stdout with length-prefixed pickles; every wait has a
  timeout and any failure makes the caller fall back to the in-process path (reward.ReportReward).
"""
from __future__ import annotations

import os
import pickle
import select
import struct
import subprocess
import sys
import time

import numpy as np


# ------------------------------------------------------------------------------------------------ sections -> strings
_NO_BATCH_DECODE = set()        # ids of tokenizer objects whose batch path disagreed with tokenizer.decode once (kept on the per-sequence path)


def _stock_fast_decode(tokenizer):
    """The tokenizer's decode is transformers' own fast-tokenizer implementation (a subclass that overrides decode / _decode keeps its own path)."""
    try:
        import transformers
        base = transformers.PreTrainedTokenizerFast
    except Exception:
        return False
    cls = type(tokenizer)
    return isinstance(tokenizer, base) and getattr(cls, "_decode", None) is getattr(base, "_decode", None) and cls.decode is base.decode


def decode_many(tokenizer, sequences):
    """[tokenizer.decode(s, skip_special_tokens=True) for s in sequences] -- the strings the reference's per-section calls produce -- through the
    batch entry point of the Rust tokenizer when `tokenizer` is a stock fast tokenizer (PreTrainedTokenizerFast._decode = backend decode +
    clean_up_tokenization where the tokenizer asks for it: restated here over `decode_batch`), through the object's own `decode_many` if it has
    one, else one by one. The restatement is CHECKED on every call: the first sequence is also decoded by tokenizer.decode itself, and a tokenizer
    whose result differs (another transformers version with other post-processing) stays on the per-sequence path from then on."""
    if not sequences:
        return []
    own = getattr(tokenizer, "decode_many", None)
    if own is not None:
        return own(sequences)
    backend = getattr(tokenizer, "backend_tokenizer", None)
    if backend is not None and hasattr(backend, "decode_batch") and id(tokenizer) not in _NO_BATCH_DECODE and _stock_fast_decode(tokenizer):
        texts = backend.decode_batch(sequences, skip_special_tokens=True)
        if getattr(tokenizer, "clean_up_tokenization_spaces", False):
            # transformers >= 5 skips the WordPiece-style clean-up for BPE models unless explicitly told otherwise (tokenization_utils_tokenizers._decode)
            bpe = type(backend.model).__name__ == "BPE"
            if not bpe or getattr(tokenizer, "clean_up_tokenization_spaces_for_bpe_even_though_it_will_corrupt_output", False):
                texts = [tokenizer.clean_up_tokenization(t) for t in texts]
        if texts[0] == tokenizer.decode(sequences[0], skip_special_tokens=True):
            return texts
        _NO_BATCH_DECODE.add(id(tokenizer))
    return [tokenizer.decode(s, skip_special_tokens=True) for s in sequences]


def split_and_decode(ids, special_token_ids, tokenizer):
    """ids: numpy int [rows, seq_len]. -> tuple (one list of `rows` strings per separator), as the reference's split_and_decode_sections."""
    ids = np.asarray(ids)
    n_rows, seq_len = ids.shape
    cols = []
    for k in special_token_ids:
        c = (ids == k).argmax(axis=1)
        c[c == 0] = seq_len                                   # quirk Q9: found at column 0 -- or not at all -- means "to the end of the row"
        cols.append(c.tolist())
    pieces = []
    for r in range(n_rows):
        prev_col = 0
        row = ids[r]
        for j in range(len(special_token_ids)):
            if prev_col >= seq_len:
                pieces.append(None)                           # -> "" (the reference appends an empty string without decoding)
                continue
            col = cols[j][r]
            pieces.append(row[prev_col:col].tolist())
            prev_col = col
    texts = iter(decode_many(tokenizer, [p for p in pieces if p is not None]))
    flat = ["" if p is None else next(texts) for p in pieces]
    ns = len(special_token_ids)
    return tuple([flat[r * ns + j] for r in range(n_rows)] for j in range(ns))


class FoldedVocabTokenizer:
    """A small-vocabulary fixture tokenizer in front of a model that emits ids up to its own (larger) vocabulary size: ids >= `first` are folded
    into [first, len(tokenizer)) before decoding, so that every generated token becomes text (benchmarks / tests with random-init models and the
    400-entry byte-BPE of tests/golden; picklable, so it can travel to a StringWorker)."""

    def __init__(self, tok, first: int = 12):
        self.tok, self.n, self.first = tok, len(tok), first

    def _fold(self, ids):
        f, n = self.first, self.n
        return [i if i < f else f + (i - f) % (n - f) for i in ids]

    def decode(self, ids, skip_special_tokens=True):
        return self.tok.decode(self._fold([int(i) for i in ids]), skip_special_tokens=skip_special_tokens)

    def decode_many(self, sequences):
        """(this fixture tokenizer does no clean-up of tokenisation spaces: the backend's strings are decode()'s strings)"""
        return self.tok.backend_tokenizer.decode_batch([self._fold(s_) for s_ in sequences], skip_special_tokens=True)


def report_tokens(id_matrices, special_token_ids, tokenizer, reward_tokenizer, encode_kw):
    """The CPU part of an SCST reward for the rows of several id matrices (in order): ids -> sections -> f"{findings} {impression}" -> reward
    tokenizer (numpy tensors). -> (input_ids [rows, R], attention_mask [rows, R], findings [rows], impression [rows])."""
    findings, impression = [], []
    for ids in id_matrices:
        _, f, i = split_and_decode(ids, special_token_ids, tokenizer)
        findings += list(f)
        impression += list(i)
    texts = [f"{i} {j}" for i, j in zip(findings, impression)]
    bep = getattr(reward_tokenizer, "batch_encode_plus", None) if "batch_encode_plus" in dir(type(reward_tokenizer)) else None
    tok = bep(batch_text_or_text_pairs=texts, **encode_kw) if bep is not None else reward_tokenizer(texts, **encode_kw)
    return np.asarray(tok["input_ids"]), np.asarray(tok["attention_mask"]), findings, impression


def report_pair_tokens(sampled, greedy, special_token_ids, tokenizer, reward_tokenizer, encode_kw):
    """Both halves of one SCST reward in one go. -> (input_ids [2B, R], attention_mask [2B, R], greedy findings, greedy impression)."""
    ids, mask, f, i = report_tokens([sampled, greedy], special_token_ids, tokenizer, reward_tokenizer, encode_kw)
    n = len(f) - np.asarray(greedy).shape[0]
    return ids, mask, f[n:], i[n:]


def pad_and_stack(parts, pad_id):
    """[(ids [b_k, L_k], mask [b_k, L_k])] -> (ids, mask) [sum b_k, max L_k]: what padding="longest" over ALL rows would have produced."""
    L = max(p[0].shape[1] for p in parts)
    ids = np.concatenate([np.pad(p[0], ((0, 0), (0, L - p[0].shape[1])), constant_values=pad_id) for p in parts], 0)
    mask = np.concatenate([np.pad(p[1], ((0, 0), (0, L - p[1].shape[1])), constant_values=0) for p in parts], 0)
    return ids, mask


# ------------------------------------------------------------------------------------------------ the child process
def _send(f, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    f.write(struct.pack("<Q", len(data)))
    f.write(data)
    f.flush()


def _recv_exact(fd, n, timeout):
    buf = bytearray()
    while len(buf) < n:
        if timeout is not None:
            r, _, _ = select.select([fd], [], [], timeout)
            if not r:
                raise TimeoutError("string worker did not answer in time")
        chunk = os.read(fd, n - len(buf))
        if not chunk:
            raise EOFError("string worker closed its pipe")
        buf += chunk
    return bytes(buf)


def _recv(fd, timeout=None):
    (n,) = struct.unpack("<Q", _recv_exact(fd, 8, timeout))
    return pickle.loads(_recv_exact(fd, n, timeout))


def _worker_main():
    """`python -m cxrmate_amd.strings`: first message = (tokenizer, reward_tokenizer, special_token_ids, encode_kw); then tuples of id matrices."""
    fin, fout = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr                                       # nothing but the protocol may reach the pipe
    fd = fin.fileno()
    try:
        tokenizer, reward_tokenizer, special, encode_kw = _recv(fd)
        _send(fout, ("ready", os.getpid()))
        while True:
            msg = _recv(fd)
            if msg is None:
                return
            seq, mats = msg                                        # every answer carries the sequence number of the request it belongs to
            try:
                _send(fout, ("ok", seq) + report_tokens(list(mats), special, tokenizer, reward_tokenizer, encode_kw))
            except Exception as e:                                 # the parent falls back to its in-process path
                _send(fout, ("error", seq, repr(e)))
    except EOFError:
        return


class StringWorker:
    """Parent-side handle of the child process. submit() is asynchronous, result() waits with a timeout; after any failure `alive` is False and
    the caller uses its in-process path. Requests carry a sequence number and result() returns only the answer to the LAST submitted request:
    a step that failed between submit() and result() (an exception in the caller, an interrupt) leaves an unread answer in the pipe, and the next
    step must not take it for its own -- answers with an older number are read and dropped."""

    def __init__(self, tokenizer, reward_tokenizer, special_token_ids, encode_kw, start_timeout: float = 120.0):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), TOKENIZERS_PARALLELISM="true")
        self.alive = False
        self.pending = False
        self.seq = 0                                              # number of the last submitted request
        self.dropped = 0                                          # stale answers discarded (tests look at it)
        self.proc = None
        try:
            self.proc = subprocess.Popen([sys.executable, "-m", "cxrmate_amd.strings"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env, cwd=root)
            _send(self.proc.stdin, (tokenizer, reward_tokenizer, list(special_token_ids), dict(encode_kw)))
            tag, _ = _recv(self.proc.stdout.fileno(), start_timeout)
            self.alive = tag == "ready"
        except Exception:                                         # OSError from fork/exec included: the caller degrades to its in-process path
            self.close()

    def submit(self, *id_matrices):
        if not self.alive:
            return False
        try:
            self.seq += 1
            _send(self.proc.stdin, (self.seq, tuple(np.ascontiguousarray(m) for m in id_matrices)))
            self.pending = True
            return True
        except Exception:
            self.close()
            return False

    def result(self, timeout: float = 30.0):
        """-> (input_ids, attention_mask, findings, impression) of the rows of the LAST submit(), or None (the caller falls back)."""
        if not (self.alive and self.pending):
            return None
        self.pending = False
        try:
            deadline = time.monotonic() + timeout
            while True:
                msg = _recv(self.proc.stdout.fileno(), max(deadline - time.monotonic(), 0.0))
                if msg[1] == self.seq:
                    break
                self.dropped += 1                                  # the answer to a request whose step never came back for it
            if msg[0] == "ok":
                return msg[2:]
            return None
        except Exception:
            self.close()
            return None

    def close(self):
        self.alive = False
        p, self.proc = getattr(self, "proc", None), None
        if p is not None:
            try:
                p.stdin.close()
            except Exception:
                pass
            try:
                p.terminate()
                p.wait(timeout=5)
            except Exception:
                try:
                    p.kill()
                except Exception:
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


if __name__ == "__main__":
    _worker_main()
