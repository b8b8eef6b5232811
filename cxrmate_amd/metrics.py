"""CXR-BERT similarity metric on MI355X -- the evaluation-time consumer of the reward kernels (reference tools/metrics/cxr_bert.py:10-184).

Behaviour kept from the reference's torchmetrics `CXRBERT(split, ckpt_dir, mbatch_size, exp_dir, accumulate_over_dicoms)`:
  * `update(predictions, labels, study_ids, dicom_ids=None)` checks its arguments the way the reference does and collects the reports;
  * `compute(epoch)` scores every (prediction, label) pair by the cosine similarity of the projected CLS embeddings, merges the rows of all
    ranks, keeps one row per DICOM / study (DDP pads the last batch with repeats), writes
    `<exp_dir>/cxr_bert/<split>_epoch-<epoch>_scores_<time>.csv` on rank 0 and returns the mean similarity, averaged per study first when
    scores were accumulated per DICOM.

How it is computed is this build's own: the state is columnar (four parallel lists), all texts are tokenised in ONE call per side, the rows are
embedded longest-first in mini-batches of `mbatch_size` (so a batch pads to its own longest report, not to the epoch's), each distinct label text
is embedded once however many predictions share it, the cosine runs on the device over the whole epoch at once, and de-duplication /
per-study averaging are numpy index operations. The two BERT-base forwards run on the same HIP kernels as the SCST reward
(`reward.CXRBERTReward.embed_ids`). torchmetrics / pandas are not dependencies. The CXR-BERT weights and tokenizer are not available offline
(DESIGN.md section 2): pass `reward=CXRBERTReward(device, tokenizer=..., state_dict=...)`.
"""
from __future__ import annotations

import csv
import os
import time

import numpy as np
import torch


def _is_dist():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


class CXRBERT:
    def __init__(self, split, ckpt_dir, mbatch_size, exp_dir, accumulate_over_dicoms, reward=None, device=None):
        self.split, self.ckpt_dir, self.mbatch_size, self.exp_dir = split, ckpt_dir, int(mbatch_size), exp_dir
        self.accumulate_over_dicoms = bool(accumulate_over_dicoms)
        self.reward = reward
        self.device = torch.device(device) if device is not None else (reward.device if reward is not None else None)
        self.save_dir = os.path.join(self.exp_dir, "cxr_bert")
        os.makedirs(self.save_dir, exist_ok=True)
        self.reset()

    # ------------------------------------------------------------------------------------------ state
    def reset(self):
        self._pred, self._label, self._study, self._dicom = [], [], [], []

    @property
    def reports(self):
        """The collected reports in the reference's record form (its torchmetrics state of the same name)."""
        keys = ("prediction", "label", "study_id") + (("dicom_id",) if self.accumulate_over_dicoms else ())
        cols = (self._pred, self._label, self._study) + ((self._dicom,) if self.accumulate_over_dicoms else ())
        return [dict(zip(keys, row)) for row in zip(*cols)]

    @staticmethod
    def mini_batch(iterable, mbatch_size=1):
        """Consecutive slices of at most `mbatch_size` items (kept for callers of the reference's helper of the same name)."""
        return (iterable[lo:lo + mbatch_size] for lo in range(0, len(iterable), mbatch_size))

    def update(self, predictions, labels, study_ids, dicom_ids=None):
        assert isinstance(predictions, list), '"predictions" must be a list of strings.'
        assert all(isinstance(p, str) for p in predictions), 'Each element of "predictions" must be a string.'
        assert isinstance(labels, list), '"labels" must be a list of lists, where each sub-list has a multiple strings.'
        assert all(isinstance(group, list) for group in labels), 'Each element of "labels" must be a list of strings.'
        assert all(isinstance(text, str) for group in labels for text in group), 'each sub-list must have one or more strings.'
        n = min(len(predictions), len(labels), len(study_ids)) if not self.accumulate_over_dicoms else \
            min(len(predictions), len(labels), len(study_ids), len(dicom_ids))
        self._pred += predictions[:n]
        self._label += labels[:n]
        self._study += list(study_ids[:n])
        if self.accumulate_over_dicoms:
            self._dicom += list(dicom_ids[:n])

    # ------------------------------------------------------------------------------------------ scoring
    def _embed_all(self, texts):
        """fp32 [len(texts), 128]: one tokenizer call, longest-first mini-batches (each pads to its own longest row), original order restored."""
        rw = self.reward
        kw = dict(add_special_tokens=True, padding=False, truncation=True, max_length=rw.config.max_position_embeddings)
        enc = rw.tokenizer(texts, **kw)["input_ids"]
        order = sorted(range(len(enc)), key=lambda i: -len(enc[i]))
        pad_id = getattr(rw.tokenizer, "pad_token_id", 0) or 0
        out = torch.empty((len(texts), rw.config.cls_projection_size), dtype=torch.float32, device=rw.device)
        for lo in range(0, len(order), self.mbatch_size):
            idx = order[lo:lo + self.mbatch_size]
            width = len(enc[idx[0]])
            ids = torch.full((len(idx), width), pad_id, dtype=torch.int64)
            mask = torch.zeros((len(idx), width), dtype=torch.int64)
            for r, i in enumerate(idx):
                ids[r, :len(enc[i])] = torch.tensor(enc[i], dtype=torch.int64)
                mask[r, :len(enc[i])] = 1
            out[torch.tensor(idx, device=rw.device)] = rw.embed_ids(ids.pin_memory().to(rw.device, non_blocking=True),
                                                                    mask.pin_memory().to(rw.device, non_blocking=True))
        return out

    def compute(self, epoch):
        from . import ops
        if self.reward is None or self.reward.tokenizer is None:
            raise RuntimeError("CXRBERT metric needs a CXRBERTReward with its tokenizer (CXR-BERT weights + tokenizer are not available offline): "
                               "pass reward=CXRBERTReward(device, tokenizer=..., state_dict=...)")
        for group in self._label:                              # following COCO the label of a report is a one-element list (cxr_bert.py:97-99)
            assert len(group) == 1
        label_texts = [group[0] for group in self._label]
        uniq = {}
        label_row = [uniq.setdefault(t, len(uniq)) for t in label_texts]        # every distinct label text is embedded once
        with torch.no_grad():
            pred_emb = self._embed_all(self._pred)
            lab_emb = self._embed_all(list(uniq))[torch.tensor(label_row, device=self.reward.device)]
            sim = ops.cosine_rows(pred_emb, lab_emb).float().cpu().numpy()
        table = {"study_id": np.asarray(self._study, dtype=object), "similarity": sim.astype(np.float64)}
        if self.accumulate_over_dicoms:
            table["dicom_id"] = np.asarray(self._dicom, dtype=object)
        if _is_dist():                                          # rows of all ranks, rank order (reference: all_gather_object of the row dicts)
            parts = [None] * torch.distributed.get_world_size()
            torch.distributed.all_gather_object(parts, table)
            table = {k: np.concatenate([p[k] for p in parts]) for k in table}
        key = "dicom_id" if self.accumulate_over_dicoms else "study_id"
        _, first = np.unique(table[key].astype(str), return_index=True)          # one row per DICOM / study: DDP pads the last batch with repeats
        keep = np.sort(first)
        table = {k: v[keep] for k, v in table.items()}
        if not _is_dist() or torch.distributed.get_rank() == 0:
            cols = (["dicom_id"] if self.accumulate_over_dicoms else []) + ["study_id", "similarity"]
            path = os.path.join(self.save_dir, f'{self.split}_epoch-{epoch}_scores_{time.strftime("%d-%m-%Y_%H-%M-%S")}.csv')
            with open(path, "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(cols)
                w.writerows(zip(*(table[c] for c in cols)))
        scores = table["similarity"]
        if self.accumulate_over_dicoms:                         # mean over the DICOMs of a study first
            _, inv = np.unique(table["study_id"].astype(str), return_inverse=True)
            scores = np.bincount(inv, weights=scores) / np.bincount(inv)
        return float(scores.mean())
