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
  * conv stack / upsample net / incremental loop / optimiser: the reference
    needs TensorFlow 1.x (``tf.layers``, ``tf.while_loop`` ...), which is not
    installable here (no network) and the reference ships no golden vectors or
    numeric tests => **parity unpinned** for those rows; they are pinned only by
    internal invariants (batch == incremental, causality, NN-init upsample ==
    repeat, autograd == finite differences) and by frozen oracle outputs.
"""
