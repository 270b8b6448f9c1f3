"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A plain-PyTorch (fp32, CPU) restatement of the hot path of zhouyuchenzyccccc/Multimodal-Sentiment-Aanalysis
(MML_ZYC/MultimodalModel.py fusion head, the CE/step sequence of Trainer.py) plus the BERT-base and ResNet-50
encoders that fill its encoder slot. Every function cites the reference file:line (or the public architecture
it restates, for the two encoders the reference does not contain).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this package, and
only as the checker / reported baseline — never as the thing measured or shipped. The product path
(`multimodal_sentiment_aanalysis_amd/`) must not import it and has no CPU fallback.

Pinning status:
  * fusion head (A1-A6): pinned — checked against the reference's own nn.Modules imported from
    /root/reference in this container (tests/golden/make_golden.py -> tests/golden/*.npz).
  * BERT-base (E1): not in the reference; pinned against transformers==5.15.0 BertModel (config-only, random
    init) by the same script. For the reference itself: "parity unpinned".
  * ResNet-50 v1.5 (E2): not in the reference (for the reference itself: "parity unpinned"); pinned against
    transformers==5.15.0 ResNetModel (config-only, downsample_in_bottleneck=False = v1.5, run in float64) by the same
    script: a mini net with stored weights (outputs, every gradient, BN running statistics, eval mode) and full
    ResNet-50 with seed-regenerated weights (tests/golden/e2_resnet_*.npz; the float64 oracle reproduces them to
    1e-12, tests/test_oracle_golden.py).
"""
