"""Token-indexing helpers with the reference's names, argument meaning and return conventions
(modules/transformers/longitudinal_model/modelling_longitudinal.py:297-513; identical copies in single/multi).
The integer work runs in the HIP kernels of csrc/misc.hip (bit-exact targets); string <-> id conversion stays in the
HuggingFace `tokenizers` library exactly as in the reference (it is part of the boundary)."""
from __future__ import annotations

import torch

from . import ops
from .strings import _NO_BATCH_DECODE, _stock_fast_decode, decode_many, split_and_decode  # noqa: F401  (torch-free home of the string work; re-exported)


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
        separator, and the B x len(special_token_ids) sections are decoded by ONE call into the `tokenizers` library (strings.decode_many) instead of
        one Python-level tokenizer.decode per section."""
        return split_and_decode(token_ids.detach().to("cpu").numpy(), special_token_ids, tokenizer)
