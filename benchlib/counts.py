"""Algorithmic work of one MPC solve and the two rooflines bench.py reports."""
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
F32_MFMA_PEAK_TF = 157.3     # dense f32-input MFMA peak (= f32 vector peak), same guide
F32_SCALAR_VALU_PEAK_TF = 78.6   # what the vector ALUs reach WITHOUT packed-f32 instructions (one fma per lane per cycle: 256 CUs x 4 SIMDs x 16 lanes x 2 flops x 2.4 GHz);
                                 # the 157.3 figure needs v_pk_fma_f32 throughout, which this kernel measured as a loss (csrc/Makefile)
F16_MFMA_PEAK_TF = 2500.0    # dense f16 / bf16 matrix peak, same guide


def algorithmic_counts(cfg, n_it, n_ls):
    """SURVEY.md §8(d) per-solve algorithmic bytes and flops (n_w = 6 noisy dims, f32)."""
    P, H, m = cfg.num_particles, cfg.horizon, cfg.num_motors
    nw = 6
    b_grad = 4 * (P * H * nw + 2 * P * (H + 1) * 13 + P * H * nw + 2 * H * m + (H + 1) * 13)
    b_ls = 4 * (P * H * nw + H * m + (H + 1) * 13)
    w_bytes = 4 * 2120
    # init-cost rollout and final mean-trajectory rollout are forward-only passes too
    bytes_solve = n_it * b_grad + (n_ls + 2) * b_ls + w_bytes
    f_step = 2 * ((6 + m) * 32 + 32 * 32 + 32 * 6) + 2 * (6 * 32 + 32 * 1)   # drift + density nets, forward
    flops_solve = f_step * P * H * (2 * n_it + n_ls + 2)
    return bytes_solve, flops_solve, b_grad, b_ls


def checkpoint_bytes(cfg, n_it):
    """Implementation stream on top of the algorithmic bytes: the gradient's forward sweep checkpoints the
    layer-2 activations + 5 step scalars per particle-step (1280 floats per 32-particle group and step),
    written once and read once per gradient evaluation."""
    G = (cfg.num_particles + 31) // 32
    return int(n_it * 2 * G * cfg.horizon * 1280 * 4)


def roofline_of(cfg, B, k_ms, n_grad, n_fwd):
    bytes_solve, flops_solve, b_grad, b_ls = algorithmic_counts(cfg, n_grad, n_fwd - 2)
    ach_gbs = bytes_solve * B / (k_ms * 1e-3) / 1e9
    ach_tf = flops_solve * B / (k_ms * 1e-3) / 1e12
    return ach_tf, ach_gbs, bytes_solve, b_grad, b_ls


def f16_contraction_flops(cfg, n_grad, n_fwd):
    """flops of the contractions the f16 mode issues on v_mfma_f32_32x32x16_f16, per solve"""
    return (2 * (6 * 64 + 32 * 32) * (n_grad + n_fwd) + 2 * (6 * 64) * n_grad) * cfg.num_particles * cfg.horizon
