"""CXR-BERT similarity metric on MI355X -- the evaluation-time twin of the SCST reward (reference tools/metrics/cxr_bert.py:10-184).

Same surface as the reference's torchmetrics `CXRBERT(split, ckpt_dir, mbatch_size, exp_dir, accumulate_over_dicoms)`:
`update(predictions, labels, study_ids, dicom_ids=None)` collects reports (same argument checks, :66-76), `compute(epoch)` embeds
predictions and labels in mini-batches (:88-131), takes the cosine similarity of the projected CLS embeddings (:128-131), gathers the rows
over ranks (:144-150), drops DDP duplicates (:154-156), writes `<exp_dir>/cxr_bert/<split>_epoch-<epoch>_scores_<time>.csv` on rank 0
(:159-171), averages over the DICOMs of a study when asked (:175-176) and returns the mean similarity (:178).

The two BERT-base forwards run on the same HIP kernels as the reward (`reward.CXRBERTReward.embed_ids`: bidirectional flash attention,
fused GEMM epilogues, no MLM head). torchmetrics is not a dependency: the state is a plain list, `reset()` clears it. The CXR-BERT weights /
tokenizer are not available offline: pass `reward=CXRBERTReward(device, tokenizer=..., state_dict=...)`.
"""
from __future__ import annotations

import os
import time
from pathlib import Path

import torch


class CXRBERT:
    def __init__(self, split, ckpt_dir, mbatch_size, exp_dir, accumulate_over_dicoms, reward=None, device=None):
        self.split = split
        self.ckpt_dir = ckpt_dir
        self.mbatch_size = mbatch_size
        self.exp_dir = exp_dir
        self.accumulate_over_dicoms = accumulate_over_dicoms
        self.reports = []
        self.reward = reward
        self.device = torch.device(device) if device is not None else (reward.device if reward is not None else None)
        self.save_dir = os.path.join(self.exp_dir, "cxr_bert")
        Path(self.save_dir).mkdir(parents=True, exist_ok=True)

    @staticmethod
    def mini_batch(iterable, mbatch_size=1):
        length = len(iterable)
        for i in range(0, length, mbatch_size):
            yield iterable[i:min(i + mbatch_size, length)]

    def reset(self):
        self.reports = []

    def update(self, predictions, labels, study_ids, dicom_ids=None):
        assert isinstance(predictions, list), '"predictions" must be a list of strings.'
        assert all(isinstance(i, str) for i in predictions), 'Each element of "predictions" must be a string.'
        assert isinstance(labels, list), '"labels" must be a list of lists, where each sub-list has a multiple strings.'
        assert all(isinstance(i, list) for i in labels), 'Each element of "labels" must be a list of strings.'
        assert all(isinstance(j, str) for i in labels for j in i), 'each sub-list must have one or more strings.'
        if self.accumulate_over_dicoms:
            for (i_1, i_2, i_3, i_4) in zip(predictions, labels, study_ids, dicom_ids):
                self.reports.append({"prediction": i_1, "label": i_2, "study_id": i_3, "dicom_id": i_4})
        else:
            for (i_1, i_2, i_3) in zip(predictions, labels, study_ids):
                self.reports.append({"prediction": i_1, "label": i_2, "study_id": i_3})

    def compute(self, epoch):
        import pandas as pd
        if self.reward is None:
            raise RuntimeError("CXRBERT metric needs a CXRBERTReward (CXR-BERT weights + tokenizer are not available offline): pass reward=...")
        rows = []
        for i in self.mini_batch(self.reports, self.mbatch_size):
            y_hat = [j["prediction"] for j in i]
            y = [j["label"] for j in i]
            study_ids = [j["study_id"] for j in i]
            if self.accumulate_over_dicoms:
                dicom_ids = [j["dicom_id"] for j in i]
            for j in y:                                      # following COCO, the labels are contained in a nested list (cxr_bert.py:97-99)
                assert len(j) == 1
            y = [j[0] for j in y]
            with torch.no_grad():
                sim = self.reward.similarity(y_hat, y)
            if self.accumulate_over_dicoms:
                rows.extend({"dicom_id": a, "study_id": b, "similarity": c} for a, b, c in zip(dicom_ids, study_ids, sim.tolist()))
            else:
                rows.extend({"study_id": a, "similarity": b} for a, b in zip(study_ids, sim.tolist()))
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            gathered = [None] * torch.distributed.get_world_size()
            torch.distributed.all_gather_object(gathered, rows)
            rows = [j for i in gathered for j in i]
        cxrbert = pd.DataFrame(rows)
        key = "dicom_id" if self.accumulate_over_dicoms else "study_id"
        cxrbert = cxrbert.drop_duplicates(subset=[key])       # duplicates caused by DDP padding
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()) or torch.distributed.get_rank() == 0:
            cxrbert.to_csv(os.path.join(self.save_dir, f'{self.split}_epoch-{epoch}_scores_{time.strftime("%d-%m-%Y_%H-%M-%S")}.csv'), index=False)
        if self.accumulate_over_dicoms:
            cxrbert = cxrbert.drop(["dicom_id"], axis=1).groupby("study_id", as_index=False).mean()
        return cxrbert.similarity.mean()
