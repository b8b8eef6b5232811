"""CPU oracle for the CXRMate hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package. The product
(`cxrmate_amd/`) never imports it and fails loudly when its HIP library is missing.

What it is: a plain fp32 PyTorch/numpy *restatement* of the arithmetic the reference executes on this path. The
reference itself is pure Python and delegates all math to third-party packages that are absent from /root/reference:

  * `transformers` (reference pins `>=4.41.2`, requirements.txt:2; installed here: 5.15.0)
        CvT   -> transformers/models/cvt/modeling_cvt.py        (cited below as TF5:cvt:LINE)
        BERT  -> transformers/models/bert/modeling_bert.py      (TF5:bert:LINE)
        generate loop / logits warpers -> transformers/generation/utils.py, logits_process.py (TF5:gen:LINE)
  * `peft` (unpinned, requirements.txt:3; NOT installed) -- LoRA on decoder self-attention query/key
        (reference modules/transformers/longitudinal_model/modelling_longitudinal.py:163-170)
  * CXR-BERT remote code `microsoft/BiomedVLP-CXR-BERT-specialized` (unpinned; unreachable offline) -- call sites
        tools/rewards/cxrbert.py:15-17,42-47. The oracle implements the declared stand-in of SURVEY.md 8(c):
        BERT-base + Linear(768,128) -> GELU -> LayerNorm(128) -> Linear(128,128) on the CLS state.

Pinning status
  * encoder / decoder / token helpers / greedy / top-k scores / REINFORCE loss: pinned against outputs of the
    reference itself, produced in the build container by importing /root/reference (script
    tests/golden/make_golden.py, transformers==5.15.0 + the two-function generate adapter of SURVEY.md 3.3) and
    committed under tests/golden/*.npz|json. tests/test_oracle_golden.py checks this package against them.
  * LoRA: pinned against the same import with the ~40-line peft stub described in SURVEY.md A.2.
  * CXR-BERT reward: **parity unpinned** (no reference source offline). The bidirectional BERT trunk is pinned
    against transformers' BertModel; the projection head is an assumption.
The reference holds no tests, golden vectors or fixtures of its own for this path.
"""
