"""The Tacotron-2-paper configuration of the reference (paper_hparams.py): copy this file over hparams.py,
or import ``hparams`` from here.  Expressed as the delta against hparams.py's defaults."""
import math

import hparams as _base

PAPER_OVERRIDES = dict(
    max_mel_frames=1000, trim_top_db=45, preemphasize=False, fmin=75, predict_linear=False,
    legacy=False, residual_legacy=False,
    log_scale_min_gauss=float(math.log(9.1188196e-4)), cdf_loss=True,
    out_channels=10 * 3, layers=24, stacks=4, residual_channels=256, gate_channels=512, skip_out_channels=256,
    upsample_type='2D', upsample_scales=[5, 5, 11], NN_scaler=0.1,
    tacotron_decay_steps=24500, tacotron_final_learning_rate=1e-5, tacotron_reg_weight=1e-7,
    wavenet_learning_rate=1e-4,
)

hparams = _base._build(PAPER_OVERRIDES)


def hparams_debug_string():
    values = hparams.values()
    hp = ['  %s: %s' % (name, values[name]) for name in sorted(values) if name != 'sentences']
    return 'Hyperparameters:\n' + '\n'.join(hp)
