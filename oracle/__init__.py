"""CPU oracle for the Long-VITA prefill hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker — never as the thing measured or shipped.  Nothing under
``long_vita_amd/`` imports it.

What it is: plain torch-CPU / numpy restatements of the reference's algorithms for every row of
SURVEY.md §8a, each function citing the reference file:line it follows
(R/ = /root/reference, M/ = R/long_vita_megatron, H/ = R/long_vita).

Pinning status ("how do we know the restatement is right"):
  * The reference ships NO tests, golden vectors or fixtures.
  * ``oracle/make_golden.py`` therefore runs the REFERENCE'S OWN PYTHON in this container
    (with stub ``megatron`` modules; the HF InternViT/projector unmodified) and commits its outputs
    under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every restatement here against
    those fixtures.  Pinned that way: zig-zag CP slice + index remap, index_of_a_in_b, RoPE table
    / apply / CP slice, RMSNorm, embedding scatter (3 forms), logits-masked linear fwd + bwd,
    pixel-shuffle, HF InternViT layer / full ViT / projector, Pillow resize / tiling / frame selection,
    get_external_inputs token surgery, and the decode-time logit-mask rule + sync_output order + block pick
    (the reference's decode loop run as CP gloo processes, decode_loop.pt), loss_func (its source executed on
    gloo ranks, loss_func.pt).
    The unfused attention math (M/core/transformer/dot_product_attention.py:151-291, run with a stand-in
    `self`, unfused_attention.pt) is pinned the same way.
  * Restated but only cross-checked (no runnable reference): decoder-layer assembly — checked against
    transformers' Qwen2 (5.x installed, reference pins >=4.48.3).
    For these rows parity is "unpinned against the reference itself".
"""
