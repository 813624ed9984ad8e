"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A plain-PyTorch fp32 restatement of the math the reference executes on the FLUX + RepText-ControlNet
denoising path. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
nothing under the product package does (tests/test_layout.py enforces that).

PARITY UNPINNED: the reference ships no tests, golden tensors or fixtures for this path and does not import
in this container (ModuleNotFoundError: diffusers — SURVEY.md §8c). The diffusers-side formulas follow
SURVEY.md Appendix A (diffusers 0.36.0, restated from knowledge); the in-repo formulas follow the cited
reference lines. What pins it instead: the closed-form known answers of SURVEY.md §8c
(tests/test_oracle_known_answers.py).
"""
