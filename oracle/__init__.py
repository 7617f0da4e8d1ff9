"""CPU oracle for the R3M hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product (r3m_amd) never
does and has no CPU path of its own. Everything here is plain PyTorch-CPU fp32 (the reference's own numerics class:
"outputs match the reference PyTorch-CPU path", BASELINE.json north_star).

Pinning: the reference has no tests and no golden vectors, and torchvision (pinned 0.8.2,
/root/reference/r3m/r3m_base.yaml:61) is not vendored. The oracle is therefore pinned against OUTPUTS OF THE REFERENCE
ITSELF run in the authoring container: tests/golden/make_golden.py imports the reference's own files by path
(oracle/by_path.py) — R3M.forward/sim, LanguageReward, Trainer.update, torch.optim.Adam — with only the torchvision ResNet
graph supplied by oracle/resnet_ref.py (restated from SURVEY.md Appendix A, checked by parameter count and state-dict key
set), and commits the resulting vectors under tests/golden/. tests/test_oracle.py checks this restatement against them.
Parity at the torchvision boundary itself stays UNPINNED (no torchvision here, pretrained checkpoints unreachable).
"""
