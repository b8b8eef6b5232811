"""CXR-BERT cosine-similarity reward on MI355X  (reference tools/rewards/cxrbert.py:9-73).

`CXRBERTReward(device)(predictions: list[str], labels: list[list[str]]) -> Tensor[B]` with the reference's argument checks.
The reference pulls `microsoft/BiomedVLP-CXR-BERT-specialized` (remote code + weights + WordPiece tokenizer) from the HF Hub
at construction time; none of that exists offline, so this class runs the declared stand-in of SURVEY.md 8(c) -- BERT-base
trunk + CLS projection head (768 -> 128 -> GELU -> LayerNorm -> 128) -- on the same HIP kernels as the decoder, random-init
unless a state dict is supplied, and takes the tokenizer as an argument. The unused MLM head of the remote class is not executed
(its output is discarded by the reference, which indexes element [2] of the tuple, quirk Q10).
Label embeddings are cached per label tuple: the reference embeds the same labels twice per SCST step.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import ops, weights
from .config import BertConfig, reward_config
from .decoder import BertEngine
from .store import ParamStore


def load_cxr_bert_checkpoint(path):
    """(tokenizer, state_dict) of a LOCAL copy of `microsoft/BiomedVLP-CXR-BERT-specialized` (what the reference's AutoModel / AutoTokenizer
    .from_pretrained(ckpt_name, cache_dir=...) pull from the Hub, tools/rewards/cxrbert.py:15-17). Expected files: the tokenizer files
    (vocab.txt / tokenizer.json ...) and `model.safetensors` or `pytorch_model.bin` with the HF key names
        bert.embeddings.{word,position,token_type}_embeddings.weight, bert.embeddings.LayerNorm.{weight,bias},
        bert.encoder.layer.{0..11}.attention.self.{query,key,value}.{weight,bias}, ...attention.output.{dense,LayerNorm}.{weight,bias},
        ...intermediate.dense.{weight,bias}, ...output.{dense,LayerNorm}.{weight,bias},
        cls_projection_head.dense_to_hidden.{weight,bias}, cls_projection_head.LayerNorm.{weight,bias}, cls_projection_head.dense_to_output.{weight,bias}
    (the MLM head `cls.predictions.*` is ignored: its output is discarded by the reference, quirk Q10). The projection-head layout is the
    stand-in assumption of SURVEY.md 8c: a checkpoint whose head has other key names fails load_state_dict loudly."""
    import transformers
    tok = transformers.AutoTokenizer.from_pretrained(path, trust_remote_code=False)
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
    keep = {k: v.float() for k, v in sd.items() if (k.startswith("bert.") or k.startswith("cls_projection_head.")) and "position_ids" not in k
            and not k.startswith("bert.pooler.")}
    return tok, keep


class CXRBERTReward:
    def __init__(self, device, tokenizer=None, config: BertConfig | None = None, state_dict=None, seed: int = 1, max_cache: int = 64,
                 ckpt_dir: str | None = None, max_length: int | None = None):
        """`CXRBERTReward(device)` is the reference's signature (tools/rewards/cxrbert.py:11). The weights / tokenizer come from, in order:
        the `tokenizer=` / `state_dict=` arguments; a local checkpoint directory `ckpt_dir=` or $CXR_BERT_DIR (load_cxr_bert_checkpoint);
        otherwise the seeded random-init stand-in without tokenizer (id-level entry points only: there is no Hub access here).
        max_length: truncation length of the tokenizer call; None = the reference's `max_position_embeddings` (512, cxrbert.py:38). bench.py sets 128
        ("decoded strings re-tokenised to a fixed R = 128 WordPiece ids") so that the string path and the synthetic-id path of the
        SCST benchmark feed the reward BERT the same number of tokens."""
        self.device = torch.device(device)
        self.config = config or reward_config()
        ckpt_dir = ckpt_dir or os.environ.get("CXR_BERT_DIR")
        if ckpt_dir and tokenizer is None and state_dict is None:
            tokenizer, state_dict = load_cxr_bert_checkpoint(ckpt_dir)
            self.config.vocab_size = int(state_dict["bert.embeddings.word_embeddings.weight"].shape[0])
        self.tokenizer = tokenizer
        self.max_length = max_length
        self.model = ParamStore(weights.bert_param_shapes(self.config, prefix=""), {}, self.device, trainable=lambda k: False)
        self.model.load_state_dict(state_dict if state_dict is not None else weights.init_reward(self.config, seed=seed))
        self.engine = BertEngine(self.model, self.config, prefix="")
        self._label_cache = {}
        self._max_cache = max_cache

    def __call__(self, predictions, labels):
        return self.reward(predictions, labels)

    # ------------------------------------------------------------------------------------------ embeddings
    @torch.no_grad()
    def embed_ids(self, input_ids, attention_mask):
        """ids/mask [B,R] -> projected CLS embedding fp32 [B,128]  (tuple element [2] of the reference call, cxrbert.py:42-47)."""
        ids, mask = self._to_device(input_ids, torch.int64, 0), self._to_device(attention_mask, torch.uint8, 1)
        hidden, _ = self.engine.forward(ids, attn_mask=mask, causal=False, lm_head=False)
        return self.engine.cls_projection(hidden)

    def _to_device(self, t, dtype, slot):
        """Host tensors (the tokenizer's output) go up through a PINNED staging buffer with an asynchronous copy: `.to(device)` from pageable memory
        makes the host wait for everything queued on the stream -- inside an SCST step that is the whole re-scoring forward, and the reward's own
        launches then start late (measured: 2 ms per step). The buffer is reused once the previous copy out of it has completed (event)."""
        if t.is_cuda:
            return t.to(device=self.device, dtype=dtype).contiguous()
        if self.device.type != "cuda":
            return t.to(dtype=dtype).contiguous()
        t = t.to(dtype=dtype).contiguous()
        st = self.__dict__.setdefault("_staging", {})
        buf, ev = st.get(slot, (None, None))
        if buf is None or buf.numel() < t.numel() or buf.dtype != dtype:
            buf = torch.empty(max(t.numel(), 1 << 14), dtype=dtype, pin_memory=True)
            ev = torch.cuda.Event()
        else:
            ev.synchronize()
        view = buf[: t.numel()].view(t.shape)
        view.copy_(t)
        out = view.to(self.device, non_blocking=True)
        ev.record()
        st[slot] = (buf, ev)
        return out

    def encode_kw(self, return_tensors="pt"):
        """Arguments of the reference's tokenizer call (tools/rewards/cxrbert.py:33-40)."""
        return dict(add_special_tokens=True, padding="longest", return_tensors=return_tensors, truncation=True,
                    max_length=self.max_length or self.config.max_position_embeddings)

    def label_embeddings(self, labels):
        """Projected CLS embeddings of the label strings, cached per label tuple (the reference embeds the same labels twice per SCST step)."""
        flat = tuple(j for i in labels for j in i)
        lab = self._label_cache.get(flat)
        if lab is None:
            lab = self._encode(list(flat))
            self._remember(flat, lab)
        return lab

    def _remember(self, flat, lab):
        if len(self._label_cache) >= self._max_cache:
            self._label_cache.clear()
        self._label_cache[flat] = lab

    def _tokenize(self, texts):
        # reference: tokenizer.batch_encode_plus(batch_text_or_text_pairs=...) (cxrbert.py:33-40); transformers 5 dropped that
        # spelling in favour of __call__ -- same arguments, same result
        kw = self.encode_kw()
        bep = getattr(self.tokenizer, "batch_encode_plus", None) if "batch_encode_plus" in dir(type(self.tokenizer)) else None
        return bep(batch_text_or_text_pairs=texts, **kw) if bep is not None else self.tokenizer(texts, **kw)

    def _encode(self, texts):
        t0 = time.perf_counter()
        tok = self._tokenize(texts)
        self.last_tokenize_ms = (time.perf_counter() - t0) * 1e3               # host time of the tokenizer call (benchmarks report it)
        return self.embed_ids(tok.input_ids, tok.attention_mask)

    def prepare_labels(self, labels):
        """Tokenise the label strings of the coming reward call NOW (host work, no GPU call): ReportReward does it while its string workers decode the
        generated ids, so that embed_with_labels finds the ids ready. No effect when the labels' embeddings are cached."""
        flat = tuple(j for i in labels for j in i)
        if flat in self._label_cache or self.tokenizer is None:
            return
        uniq = list(dict.fromkeys(flat))
        tok = self._tokenize(uniq)
        self.__dict__["_label_tok"] = (flat, uniq, tok.input_ids, tok.attention_mask)

    @torch.no_grad()
    def embed_with_labels(self, pred_ids, pred_mask, labels):
        """(embeddings of the prediction rows [P,128], embeddings of the flattened labels [len,128]). The reference embeds predictions and labels by two
        model calls per reward() and calls reward() twice per SCST step with the same labels (cxrbert.py:49-64; scst/gt_prompt.py:90-91,126-128). Here
        the labels of a step are embedded ONCE -- duplicates removed -- and, when they are not cached from an earlier call, IN THE SAME forward as the
        prediction rows (one 3B-row launch chain instead of a 2B-row and a B-row one: a BERT-base forward over <= 48 rows is launch-bound).
        Padding both to the common length changes nothing: padded keys are masked and only the CLS row is read."""
        flat = tuple(j for i in labels for j in i)
        lab = self._label_cache.get(flat)
        if lab is not None:
            return self.embed_ids(pred_ids, pred_mask), lab
        pre = self.__dict__.pop("_label_tok", None)
        if pre is not None and pre[0] == flat:
            uniq, lids, lmask = pre[1], pre[2], pre[3]
        else:
            uniq = list(dict.fromkeys(flat))
            tok = self._tokenize(uniq)
            lids, lmask = tok.input_ids, tok.attention_mask
        pred_ids, pred_mask, lids, lmask = (torch.as_tensor(t) for t in (pred_ids, pred_mask, lids, lmask))
        pad = getattr(self.tokenizer, "pad_token_id", None) or 0
        L = max(pred_ids.shape[1], lids.shape[1])

        def widen(t, fill):
            if t.shape[1] == L:
                return t
            out = t.new_full((t.shape[0], L), fill)
            out[:, : t.shape[1]] = t
            return out
        dev_side = pred_ids.device
        ids = torch.cat([widen(pred_ids, pad), widen(lids.to(dev_side), pad)], 0)
        mask = torch.cat([widen(pred_mask, 0), widen(lmask.to(dev_side), 0).to(pred_mask.dtype)], 0)
        emb = self.embed_ids(ids, mask)
        P = pred_ids.shape[0]
        pos = {s_: i for i, s_ in enumerate(uniq)}
        lab_u = emb[P:]
        if len(flat) == len(uniq):
            lab = lab_u
        else:
            lab = lab_u[torch.tensor([pos[s_] for s_ in flat], dtype=torch.int64, device=emb.device)]      # (row copies, no arithmetic)
        self._remember(flat, lab)
        return emb[:P], lab

    def reward(self, predictions, labels):
        assert isinstance(predictions, list), '"predictions" must be a list of strings.'
        assert all(isinstance(i, str) for i in predictions), 'Each element of "predictions" must be a string.'
        assert isinstance(labels, list), '"labels" must be a list of lists, where each sub-list has a multiple strings.'
        assert all(isinstance(i, list) for i in labels), 'Each element of "labels" must be a list of strings.'
        assert all(isinstance(j, str) for i in labels for j in i), 'each sub-list must have one or more strings.'
        if self.tokenizer is None:
            raise RuntimeError("CXRBERTReward needs the CXR-BERT tokenizer (not available offline): pass tokenizer=...")
        t0 = time.perf_counter()
        tok = self._tokenize(predictions)
        self.last_tokenize_ms = (time.perf_counter() - t0) * 1e3               # host time of the tokenizer call (benchmarks report it)
        pred, lab = self.embed_with_labels(tok.input_ids, tok.attention_mask, labels)
        return ops.cosine_rows(pred, lab)

    @torch.no_grad()
    def similarity(self, predictions, labels):
        """Cosine similarity of the projected CLS embeddings of two equally long lists of strings, no label caching
        (the evaluation metric's inner step, reference tools/metrics/cxr_bert.py:101-131)."""
        if self.tokenizer is None:
            raise RuntimeError("CXRBERTReward needs the CXR-BERT tokenizer (not available offline): pass tokenizer=...")
        return ops.cosine_rows(self._encode(list(predictions)), self._encode(list(labels)))

    @torch.no_grad()
    def reward_from_ids(self, pred_ids, pred_mask, label_ids, label_mask):
        """Tokenizer-free entry (synthetic benchmarks / parity tests)."""
        return ops.cosine_rows(self.embed_ids(pred_ids, pred_mask), self.embed_ids(label_ids, label_mask))


class ReportReward:
    """`reward_fn` for scst.scst_step(..., reward_on_host=True): the reference's string round trip inside an SCST step
    (modules/lightning_modules/longitudinal/scst/gt_prompt.py:90-91,120-128,192-197): token ids -> findings / impression strings
    (`split_and_decode_sections` with [bos, sep, eos], the previous-report prompt excluded) -> f"{findings} {impression}" -> CXR-BERT reward
    against the study's labels. Works on PINNED HOST copies of the sequences (one asynchronous copy per decode, no per-row sync); the label
    embeddings are computed once per batch (CXRBERTReward caches them per label tuple: the reference embeds them twice per step)."""

    def __init__(self, model, tokenizer, reward: CXRBERTReward, labels, bos_token_id, sep_token_id, eos_token_id, worker: bool = False):
        """worker=True: the CPU part of pair() -- ids -> strings -> reward-tokenizer ids -- runs in a child process (strings.StringWorker) between
        pair_start() and pair_finish(), i.e. WHILE this process's Python thread queues the re-scoring pass: scst.scst_step uses the two calls when
        they exist. Both tokenizers must pickle (HF fast tokenizers and strings.FoldedVocabTokenizer do). Any failure of the child -- start-up,
        time-out, exception -- falls back to the in-process path; results are the same either way."""
        self.model, self.tokenizer, self.reward = model, tokenizer, reward
        self.labels = labels                                  # [[f"{findings} {impression}"], ...] as the reference builds them (gt_prompt.py:90)
        self.special = [bos_token_id, sep_token_id, eos_token_id]

        self.last_sections = None                             # (findings, impression) strings of the most recent call
        self.workers = []
        self.worker_used = 0                                  # pairs the child processes have served (benchmarks / tests report it)
        if worker and reward.tokenizer is not None:
            from .strings import StringWorker
            # k children per half (sampled rows | greedy rows; CXR_STRING_WORKERS = 2 k children in all, default 4): the chunks are decoded and
            # re-tokenised side by side, and the parent pads them to the common length (what padding="longest" over all rows gives)
            n = max(2, int(os.environ.get("CXR_STRING_WORKERS", "4")) // 2 * 2)
            ws = []
            try:
                for _ in range(n):
                    ws.append(StringWorker(tokenizer, reward.tokenizer, self.special, reward.encode_kw(return_tensors="np")))
            except Exception:                                 # (StringWorker itself degrades on start-up failures; this is for what it cannot foresee)
                pass
            if len(ws) == n and all(w.alive for w in ws):
                self.workers = ws
            else:
                for w in ws:                                  # whichever were created: none is left running
                    w.close()

    @property
    def worker(self):
        """The first child process (None without workers): kept for callers / tests that look at one."""
        return self.workers[0] if self.workers else None

    def pair_start(self, sampled_host, greedy_host):
        """Hand both halves' ids (host tensors whose copies have LANDED) to the child processes. -> ticket for pair_finish()."""
        k = len(self.workers) // 2
        if k and all(w.alive for w in self.workers) and sampled_host.shape[0] >= k:
            chunks = np.array_split(sampled_host.numpy(), k) + np.array_split(greedy_host.numpy(), k)      # contiguous row ranges, in row order
            if all(w.submit(c) for w, c in zip(self.workers, chunks)):
                self.reward.prepare_labels(self.labels)       # this step's labels are tokenised here while the children decode the generated ids
                return ("worker", sampled_host, greedy_host)
        return ("inline", sampled_host, greedy_host)

    def pair_finish(self, ticket):
        kind, sampled_host, greedy_host = ticket
        got = [w.result() for w in self.workers] if kind == "worker" else [None]
        if any(g is None for g in got):                       # no workers, or one failed: the in-process path (same strings, same ids)
            for w in self.workers:
                w.pending = False
            if self.workers and not all(w.alive for w in self.workers):
                self.close()
            return self.pair(sampled_host, greedy_host)
        from .strings import pad_and_stack
        pad_id = self.reward.tokenizer.pad_token_id
        ids, mask = pad_and_stack([(g[0], g[1]) for g in got], 0 if pad_id is None else pad_id)
        k = len(got) // 2                                     # the greedy rows' sections, chunk by chunk
        self.last_sections = ([x for g in got[k:] for x in g[2]], [x for g in got[k:] for x in g[3]])
        self.worker_used += 1
        pe, le = self.reward.embed_with_labels(torch.from_numpy(ids), torch.from_numpy(mask), self.labels + self.labels)
        both = ops.cosine_rows(pe, le)
        B = ids.shape[0] // 2
        return both[:B], both[B:]

    def close(self):
        for w in self.workers:
            w.close()
        self.workers = []

    def pair(self, sampled_host, greedy_host):
        """(reward of the sampled rows, reward of the greedy rows) from ONE tokenizer call and ONE 2B-row CXR-BERT forward; `last_sections` holds the
        greedy rows' sections afterwards (what the generated-prompt caller writes back)."""
        t0 = time.perf_counter()
        _, fs, is_ = self.model.split_and_decode_sections(sampled_host, self.special, self.tokenizer)
        _, fg, ig = self.model.split_and_decode_sections(greedy_host, self.special, self.tokenizer)
        self.last_sections = (fg, ig)
        self.last_decode_ms = (time.perf_counter() - t0) * 1e3                  # ids -> strings, both halves (host)
        both = self.reward.reward([f"{i} {j}" for i, j in zip(list(fs) + list(fg), list(is_) + list(ig))], self.labels + self.labels)
        B = len(fs)
        return both[:B], both[B:]

    def __call__(self, sequences_host):
        _, findings, impression = self.model.split_and_decode_sections(sequences_host, self.special, self.tokenizer)
        self.last_sections = (findings, impression)
        return self.reward.reward([f"{i} {j}" for i, j in zip(findings, impression)], self.labels)
