"""CPU oracle for the WaveNet-vocoder hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU fp32 + numpy) of the arithmetic of
Rayhane-mamah/Tacotron-2's ``wavenet_vocoder/models/{wavenet,modules,mixture,
gaussian}.py`` and ``wavenet_vocoder/util.py``.  Every function cites the
reference file:line it follows.

It is the *checker*, never the product:
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it;
  * nothing under ``tacotron-2_amd/`` imports it, and the product path raises
    if the HIP library is missing (no CPU fallback).

PINNING STATUS
  * mu-law codec, MoL / Gaussian losses and samplers: pinned against the
    reference's OWN source files executed in this container (numpy code path of
    ``util.py`` directly; ``mixture.py`` / ``gaussian.py`` through a minimal
    eager stand-in for the handful of ``tf.*`` element-wise ops they call) --
    see ``oracle/gen_golden.py`` and ``tests/golden/*.npz``.
  * conv stack / upsample net (all five types) / masked training loss (add_loss) / incremental loop incl. the
    sample feedback / global conditioning / use_bias=False: pinned against the reference's OWN
    ``models/wavenet.py`` and ``models/modules.py`` EXECUTED in this container: TensorFlow 1.x is not
    installable (no network), so ``oracle/tf1_shim.py`` serves the ~70 ``tf.*`` symbols those files use with
    eager torch ops of the documented TF semantics (kernel layouts, SAME/VALID padding, channels_first/last,
    batch_to_space_nd, TensorArray/while_loop ...), and ``oracle/gen_golden_stack.py`` records
    ``WaveNet.step`` / ``.incremental`` / ``.add_loss`` outputs on 18 configurations (incl. weight normalisation and the engine-width `hip_*` ones) into
    ``tests/golden/stack_*.npz``.  What this pins is the reference's COMPOSITION of ops (order, transposes,
    paddings, splits, scalings, queue updates) -- the part a restatement can get wrong; the primitives of the
    stand-in are themselves checked against naive index loops (tests/test_oracle_golden.py).  It is not a run
    of real TensorFlow kernels: float summation order inside a conv differs at the 1e-6 level.
  * optimiser (tf.train.AdamOptimizer / ExponentialMovingAverage / exponential_decay are TF library code, not
    reference code): restated from the TF documentation, checked by closed forms => "parity unpinned" for
    that row only.
"""
