"""numpy restatement of the reference's integer/token-indexing helpers (TEST INFRASTRUCTURE; see oracle/__init__.py).

Bit-exact targets. Citations are to reference modules/transformers/longitudinal_model/modelling_longitudinal.py
(identical copies live in single_model/modelling_single.py:251-318 and multi_model/modelling_multi.py).
"""
from __future__ import annotations

import numpy as np


def token_ids_to_token_type_ids(token_ids, special_token_ids, token_type_id_sections=None):
    """modelling_longitudinal.py:297-338 (quirk Q4: section switches AFTER the separator; a separator in column 0
    or in the last column counts as "not found")."""
    ids = np.asarray(token_ids, dtype=np.int64)
    sections = list(token_type_id_sections) if token_type_id_sections is not None else list(range(len(special_token_ids) + 1))
    b, t = ids.shape
    out = np.full((b, t), sections[0], dtype=np.int64)
    for i, j in enumerate(special_token_ids):
        cols = (ids == j).astype(np.int32).argmax(axis=1) + 1          # :317,321 first occurrence, +1
        for r in range(b):
            if cols[r] != 1 and cols[r] < t:                             # :325-326
                out[r, cols[r]:] = sections[i + 1]                       # :330-336
    return out


def token_ids_to_token_type_ids_past(token_ids, special_token_ids, token_type_id_sections=None):
    """modelling_longitudinal.py:340-364: type of the NEXT position given everything but the last column."""
    ids = np.asarray(token_ids, dtype=np.int64)
    sections = list(token_type_id_sections) if token_type_id_sections is not None else list(range(len(special_token_ids) + 1))
    out = np.full((ids.shape[0], 1), sections[0], dtype=np.int64)
    ids = ids[:, :-1]                                                    # :356
    for i, j in enumerate(special_token_ids):
        exists = (ids == j).any(axis=1)                                  # :361
        out[exists, 0] = sections[i + 1]
    return out


def position_ids_from_mask(mask):
    """modelling_longitudinal.py:274-277: relu(cumsum(mask) - 1)."""
    m = np.asarray(mask, dtype=np.int64)
    return np.maximum(np.cumsum(m, axis=1) - 1, 0)


def teacher_forcing_split(input_ids, attention_mask):
    """modelling_longitudinal.py:398-409 applied to already-tokenised ids (quirk Q6: mask is shifted LEFT)."""
    ids = np.asarray(input_ids, dtype=np.int64)
    am = np.asarray(attention_mask, dtype=np.int64)
    return {"label_ids": ids[:, 1:].copy(), "decoder_input_ids": ids[:, :-1], "decoder_attention_mask": am[:, 1:]}


def force_bos_last(input_ids, attention_mask, max_len, bos_token_id):
    """modelling_longitudinal.py:503-509 (quirk Q8)."""
    ids = np.array(input_ids, dtype=np.int64, copy=True)
    am = np.asarray(attention_mask, dtype=np.int64)
    if ids.shape[1] == max_len:
        ids[:, -1] = np.where(am[:, -1] == 1, bos_token_id, ids[:, -1])
    return ids


def split_sections(token_ids, special_token_ids):
    """Index arithmetic of split_and_decode_sections (modelling_longitudinal.py:413-457) without the tokenizer:
    returns, per section j, the list of id arrays that the reference hands to tokenizer.decode (quirk Q9)."""
    ids = np.asarray(token_ids, dtype=np.int64)
    _, t = ids.shape
    sections = {k: [] for k in range(len(special_token_ids))}
    for row in ids:
        prev = 0
        for j, k in enumerate(special_token_ids):
            if prev >= t:                                                # :438-440
                sections[j].append(None)
                continue
            col = int((row == k).astype(np.int32).argmax())
            if col == 0:                                                 # :447-448
                col = t
            sections[j].append(row[prev:col].copy())
            prev = col
    return tuple(sections.values())


def strip_prepended_bos(ids, bos_token_id=1):
    """modelling_longitudinal.py:270-271 / scst/gt_prompt.py:117-118."""
    ids = np.asarray(ids)
    return ids[:, 1:] if np.all(ids[:, 0] == bos_token_id) else ids
