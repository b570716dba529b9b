"""numpy restatement of one iteration of train_gan.train_gan_one_epoch.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows train_gan.py:52-71 with the
minimal schedule that reproduces the reference's results: the
generator forward runs once and its BN buffers take the update twice; the
discriminator backward of the G phase computes only the input gradient.
"""
import numpy as np

from . import nets, ops

LAMBDA_L1 = 100.0          # train_gan.py:33
LR = 1e-4                  # train_gan.py:28-29
BETAS = (0.5, 0.999)       # train_gan.py:30,140-141


def is_param(key):
    return not (key.endswith("running_mean") or key.endswith("running_var") or key.endswith("num_batches_tracked"))


def new_adam_state(st):
    return {k: dict(m=np.zeros_like(v), v=np.zeros_like(v), t=0) for k, v in st.items() if is_param(k)}


def _adam(st, grads, opt):
    for k, g in grads.items():
        s = opt[k]
        s["t"] += 1
        st[k], s["m"], s["v"] = ops.adam_step(st[k], g, s["m"], s["v"], s["t"], LR, BETAS[0], BETAS[1])


def gan_step(g_st, d_st, g_opt, d_opt, real_a, real_b, num_downs=7, n_layers=3):
    """One batch of train_gan_one_epoch.  Mutates the state/optimizer dicts in place.

    Returns (loss_d, loss_g, extras) with extras holding fake_B and the raw
    gradients for finer-grained parity checks.
    """
    # ---- D phase (train_gan.py:55-63)
    fake_b, g_cache, g_nb = nets.generator_fwd(g_st, real_a, num_downs, True, bn_repeats=2)
    pred_real, c_real, nb1 = nets.discriminator_fwd(d_st, np.concatenate([real_a, real_b], 1), n_layers, True)
    d_st.update(nb1)
    l_real, dl_real = ops.bce_logits(pred_real, 1.0)
    pred_fake, c_fake, nb2 = nets.discriminator_fwd(d_st, np.concatenate([real_a, fake_b], 1), n_layers, True)
    d_st.update(nb2)
    l_fake, dl_fake = ops.bce_logits(pred_fake, 0.0)
    loss_d = np.float32((l_real + l_fake) * np.float32(0.5))
    gr, _ = nets.discriminator_bwd(d_st, c_real, dl_real * np.float32(0.5))
    gf, _ = nets.discriminator_bwd(d_st, c_fake, dl_fake * np.float32(0.5))
    d_grads = {k: (gr[k] + gf[k]).astype(np.float32) for k in gr}
    _adam(d_st, d_grads, d_opt)
    # ---- G phase (train_gan.py:64-71); G forward #2 == #1 bitwise, buffers already updated twice
    g_st.update(g_nb)
    pred_g, c_g, nb3 = nets.discriminator_fwd(d_st, np.concatenate([real_a, fake_b], 1), n_layers, True)
    d_st.update(nb3)
    l_gan, dl_gan = ops.bce_logits(pred_g, 1.0)
    l_l1, dl_l1 = ops.l1_loss(fake_b, real_b)
    loss_g = np.float32(l_gan + np.float32(LAMBDA_L1) * l_l1)
    _, d_in = nets.discriminator_bwd(d_st, c_g, dl_gan, need_dx=True)
    d_fake = d_in[:, real_a.shape[1]:] + np.float32(LAMBDA_L1) * dl_l1
    g_grads = nets.generator_bwd(g_st, g_cache, d_fake.astype(np.float32), num_downs)
    _adam(g_st, g_grads, g_opt)
    return loss_d, loss_g, dict(fake_b=fake_b, d_grads=d_grads, g_grads=g_grads, pred_real=pred_real,
                                pred_fake=pred_fake, pred_g=pred_g)
