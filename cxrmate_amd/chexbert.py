"""CheXbert labeller on MI355X (reference tools/chexbert.py:9-83, used by the checkpoint-selection metric
`val_report_chexbert_f1_macro`): BERT-base trunk -> CLS state -> 14 linear heads (13 observations x {blank, positive, negative,
uncertain}, `no finding` x {yes, no}) -> argmax per head.

Same call surface: `CheXbert(...)(reports: list[str]) -> LongTensor [B, 14]`; the report clean-up lines are kept verbatim (including the
two `str.replace` calls with regex-looking patterns, which are literal replacements in the reference and therefore no-ops). The trunk runs on
the bidirectional path of the decoder engine (flash attention with key padding, fused GEMM epilogues); the 14 heads are ONE GEMM over the
stacked head weights followed by a segmented argmax kernel. nn.Dropout(p) on the CLS state is identity (the reference module is in eval mode).
The checkpoint and the `bert-base-uncased` tokenizer are not available offline: pass `tokenizer=` and `state_dict=` (keys `bert.*`,
`linear_heads.{i}.{weight,bias}`; a `module.` prefix from DataParallel checkpoints is stripped as in the reference, :39-45).
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from . import ops, weights
from ._lib import LIB
from .config import BertConfig
from .decoder import BertEngine
from .store import ParamStore

HEAD_CLASSES = (4,) * 13 + (2,)


def chexbert_config() -> BertConfig:
    return BertConfig(vocab_size=30522, num_hidden_layers=12, is_decoder=False, add_cross_attention=False, cls_projection_size=0)


def chexbert_param_shapes(cfg: BertConfig):
    shapes = OrderedDict((k, v) for k, v in weights.bert_param_shapes(cfg, prefix="").items() if not k.startswith("cls."))
    d = cfg.hidden_size
    shapes["bert.pooler.dense.weight"] = (d, d)                     # present in the checkpoint, unused by the labeller (:70 takes [0])
    shapes["bert.pooler.dense.bias"] = (d,)
    for i, n in enumerate(HEAD_CLASSES):
        shapes[f"linear_heads.{i}.weight"] = (n, d)
        shapes[f"linear_heads.{i}.bias"] = (n,)
    return shapes


class CheXbert:
    def __init__(self, *args, device=None, tokenizer=None, config: BertConfig | None = None, state_dict=None, seed: int = 2, p: float = 0.1,
                 ckpt_dir=None, bert_path=None, checkpoint_path=None):
        """Two call forms:
          * the reference's `CheXbert(ckpt_dir, bert_path, checkpoint_path, device, p=0.1)` (tools/chexbert.py:10): tokenizer from the LOCAL
            directory `bert_path` (BertTokenizer files of bert-base-uncased), weights from `os.path.join(ckpt_dir, checkpoint_path)`
            (`['model_state_dict']`, a missing file raises the reference's ValueError);
          * `CheXbert(device, tokenizer=..., state_dict=...)` / `CheXbert(device)` (seeded random init) for tests and benchmarks."""
        import os
        if len(args) == 4:
            ckpt_dir, bert_path, checkpoint_path, device = args
        elif len(args) == 1:
            device = args[0]
        elif args:
            raise TypeError("CheXbert(ckpt_dir, bert_path, checkpoint_path, device, p=0.1) or CheXbert(device, tokenizer=..., state_dict=...)")
        if checkpoint_path is not None and state_dict is None:
            ckpt_path = os.path.join(ckpt_dir, checkpoint_path)
            if not os.path.isfile(ckpt_path):
                raise ValueError(f'The CheXbert checkpoint does not exist at {ckpt_dir}, please download it from: '
                                 'https://github.com/stanfordmlgroup/CheXbert#checkpoint-download.')
            state_dict = torch.load(ckpt_path, map_location="cpu")["model_state_dict"]
        if bert_path is not None and tokenizer is None:
            import transformers
            tokenizer = transformers.BertTokenizer.from_pretrained(bert_path, cache_dir=ckpt_dir)
        self.ckpt_dir = ckpt_dir
        self.device = torch.device(device)
        self.config = config or chexbert_config()
        self.tokenizer = tokenizer
        shapes = chexbert_param_shapes(self.config)
        self.model = ParamStore(shapes, {}, self.device, trainable=lambda k: False)
        if state_dict is None:
            state_dict = weights.init_state_dict(shapes, seed=seed, std=self.config.initializer_range, perturb=0.05)
        else:
            state_dict = OrderedDict((k.replace("module.bert.", "bert.").replace("module.linear_heads.", "linear_heads."), v)
                                     for k, v in state_dict.items() if "position_ids" not in k)
        self.model.load_state_dict(state_dict)
        self.engine = BertEngine(self.model, self.config, prefix="")
        n = sum(HEAD_CLASSES)
        self._n = n
        self._off = torch.tensor([0] + list(torch.cumsum(torch.tensor(HEAD_CLASSES), 0)), dtype=torch.int32, device=self.device)
        self._version = -1

    def _heads(self):
        """Stacked head weights [64 (54 used), 768] bf16 + bias fp32 [64], rebuilt when the weights change (layout plumbing)."""
        st = self.model
        st.refresh_shadow()
        if self._version != st.shadow_version:
            w = torch.zeros((64, self.config.hidden_size), dtype=torch.bfloat16, device=self.device)
            b = torch.zeros((64,), dtype=torch.float32, device=self.device)
            o = 0
            for i, n in enumerate(HEAD_CLASSES):
                w[o:o + n] = st.w16(f"linear_heads.{i}.weight")
                b[o:o + n] = st.f32(f"linear_heads.{i}.bias")
                o += n
            self._w, self._b, self._version = w, b, st.shadow_version
        return self._w, self._b

    @torch.no_grad()
    def label_ids(self, input_ids, attention_mask, token_type_ids=None):
        """ids / mask [B,R] -> class index per head, int64 [B,14] (and the stacked head logits fp32 [B,54])."""
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        mask = attention_mask.to(device=self.device, dtype=torch.uint8).contiguous()
        tt = None if token_type_ids is None else token_type_ids.to(device=self.device, dtype=torch.int64).contiguous()
        hidden, _ = self.engine.forward(ids, attn_mask=mask, token_type_ids=tt, causal=False, lm_head=False, train=False)
        w, b = self._heads()
        logits = ops.gemm_nt(hidden[:, 0, :], w, bias=b, out_f32=True)                      # [B, 64]; CLS rows are a strided view
        out = torch.empty((ids.shape[0], len(HEAD_CLASSES)), dtype=torch.int64, device=self.device)
        LIB.call("cxr_segment_argmax_f32", ops._p(logits), logits.stride(0), ops._p(self._off), len(HEAD_CLASSES), ops._p(out), ids.shape[0], ops._s())
        return out, logits[:, :self._n]

    def __call__(self, reports):
        return self.forward(reports)

    def forward(self, reports):
        if self.tokenizer is None:
            raise RuntimeError("CheXbert needs the bert-base-uncased tokenizer (not available offline): pass tokenizer=...")
        for i in range(len(reports)):                                   # reference tools/chexbert.py:53-58, verbatim
            reports[i] = reports[i].strip()
            reports[i] = reports[i].replace("\n", " ")
            reports[i] = reports[i].replace("\\s+", " ")
            reports[i] = reports[i].replace("\\s+(?=[\\.,])", "")
            reports[i] = reports[i].strip()
        tok = self.tokenizer(reports, padding="longest", return_tensors="pt", truncation=True, max_length=self.config.max_position_embeddings)
        return self.label_ids(tok["input_ids"], tok["attention_mask"], tok.get("token_type_ids"))[0]
