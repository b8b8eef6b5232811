"""Token-indexing helpers with the reference's names, argument meaning and return conventions
(modules/transformers/longitudinal_model/modelling_longitudinal.py:297-513; identical copies in single/multi).
The integer work runs in the HIP kernels of csrc/misc.hip (bit-exact targets); string <-> id conversion stays in the
HuggingFace `tokenizers` library exactly as in the reference (it is part of the boundary)."""
from __future__ import annotations

import torch

from . import ops


class TokenHelpers:
    # -------------------------------------------------------------------------------------- integer kernels
    def token_ids_to_token_type_ids(self, token_ids, special_token_ids, token_type_id_sections=None):
        """reference :297-338."""
        ids = token_ids.to(device=self.device, dtype=torch.int64).contiguous()
        return ops.token_type_ids(ids, special_token_ids, token_type_id_sections, past=False)

    def token_ids_to_token_type_ids_past(self, token_ids, special_token_ids, token_type_id_sections=None):
        """reference :340-364."""
        ids = token_ids.to(device=self.device, dtype=torch.int64).contiguous()
        return ops.token_type_ids(ids, special_token_ids, token_type_id_sections, past=True)

    def position_ids_from_mask_token(self, token_ids, mask_token_id):
        """decoder_attention_mask = (ids != mask_token_id); position = relu(cumsum(mask) - 1)  (reference :274-277)."""
        ids = token_ids.to(device=self.device, dtype=torch.int64).contiguous()
        return ops.mask_position_ids(ids, mask_token_id)

    # -------------------------------------------------------------------------------------- tokenizer plumbing
    def tokenize_report_teacher_forcing(self, findings, impression, tokenizer, max_len: int):
        """reference :366-411 (quirk Q6: the attention mask is shifted left)."""
        report = [f"{tokenizer.bos_token}{i}{tokenizer.sep_token}{j}{tokenizer.eos_token}" for i, j in zip(findings, impression)]
        tokenized = tokenizer(report, padding="longest", truncation=True, max_length=max_len + 1, return_tensors="pt",
                              return_token_type_ids=False, add_special_tokens=False).to(self.device)
        return {
            "label_ids": tokenized["input_ids"][:, 1:].detach().clone(),
            "decoder_input_ids": tokenized["input_ids"][:, :-1],
            "decoder_attention_mask": tokenized["attention_mask"][:, 1:],
        }

    def tokenize_prompt(self, previous_findings, previous_impression, tokenizer, max_len: int, add_bos_token_id: bool = False):
        """reference :459-513 (quirk Q8: BOS forced into the last column when the prompt fills max_len)."""
        previous_findings = ["[NPF]" if not i else i for i in previous_findings]
        previous_impression = ["[NPI]" if not i else i for i in previous_impression]
        sections = [f"[PMT]{i}[PMT-SEP]{j}{tokenizer.bos_token}" if add_bos_token_id else f"[PMT]{i}[PMT-SEP]{j}"
                    for i, j in zip(previous_findings, previous_impression)]
        tok = tokenizer(sections, padding="longest", truncation=True, max_length=max_len, return_tensors="pt",
                        return_token_type_ids=False, add_special_tokens=False).to(self.device)
        input_ids, attention_mask = tok.input_ids, tok.attention_mask
        if input_ids.shape[1] == max_len:
            input_ids[:, -1] = torch.where(attention_mask[:, -1] == 1, tokenizer.bos_token_id, input_ids[:, -1])
        assert input_ids.shape[1] <= max_len
        return {"input_ids": input_ids, "attention_mask": attention_mask}

    def split_and_decode_sections(self, token_ids, special_token_ids, tokenizer):
        """reference :413-457 (quirk Q9: a separator found at column 0 -- or not at all -- ends its section at the end of the row). One device->host
        copy for the whole batch instead of an .item() sync per row/section; the separator columns of all rows come from one numpy comparison per
        separator, and the B x len(special_token_ids) sections are decoded by ONE call into the `tokenizers` library (decode_many) instead of one
        Python-level tokenizer.decode per section: inside an SCST step this host work runs while the GPU re-scores (scst.scst_step)."""
        ids = token_ids.detach().to("cpu").numpy()
        n_rows, seq_len = ids.shape
        cols = []
        for k in special_token_ids:
            c = (ids == k).argmax(axis=1)
            c[c == 0] = seq_len
            cols.append(c.tolist())
        pieces = []
        for r in range(n_rows):
            prev_col = 0
            row = ids[r]
            for j in range(len(special_token_ids)):
                if prev_col >= seq_len:
                    pieces.append(None)                       # -> "" (the reference appends an empty string without decoding)
                    continue
                col = cols[j][r]
                pieces.append(row[prev_col:col].tolist())
                prev_col = col
        texts = iter(decode_many(tokenizer, [p for p in pieces if p is not None]))
        flat = ["" if p is None else next(texts) for p in pieces]
        ns = len(special_token_ids)
        return tuple([flat[r * ns + j] for r in range(n_rows)] for j in range(ns))


_NO_BATCH_DECODE = set()        # ids of tokenizer objects whose batch path disagreed with tokenizer.decode once (kept on the per-sequence path)


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


def _stock_fast_decode(tokenizer):
    """The tokenizer's decode is transformers' own fast-tokenizer implementation (a subclass that overrides decode / _decode keeps its own path)."""
    try:
        import transformers
        base = transformers.PreTrainedTokenizerFast
    except Exception:
        return False
    cls = type(tokenizer)
    return isinstance(tokenizer, base) and getattr(cls, "_decode", None) is getattr(base, "_decode", None) and cls.decode is base.decode
