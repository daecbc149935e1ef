// ctk_adam.h — the per-element update rules of the gradient-based optimizers (shared by ctk_rpgd.hip and ctk_generic.hip)
#pragma once
#include "ctk_device.h"

struct AdamK {
    float lr, b1, b2, one_m_b1, one_m_b2, eps, clip;
    int rule;   // 0: in-repo torch Adam (optimizer_rpgd.py:56-82); 1: Keras Adam (gradient_tf, bharadhwaj); 2: plain SGD (cem_naive_grad)
};

// Adam for one element (optimizer_rpgd.py:68-79 in fp32, scalars rounded to fp32 as torch does)
CTK_DEV float adam_update(const AdamK& ad, float q, float g, float& m, float& v, float bc1, float bc2, float lo, float hi) {
    if (ad.rule == 2) return fminf(fmaxf(q - ad.lr * g, lo), hi);   // optimizer_cem_naive_grad_tf.py:70-71
    m = m * ad.b1 + ad.one_m_b1 * g;
    v = v * ad.b2 + ad.one_m_b2 * (g * g);
    if (ad.rule == 1) {
        // tf.keras.optimizers.Adam (third party; published update rule): lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
        // var -= lr_t * m / (sqrt(v) + eps)   — epsilon is NOT bias-corrected, unlike the torch branch
        const float lr_t = ad.lr * sqrtf(bc2) / bc1;
        return fminf(fmaxf(q - lr_t * m / (sqrtf(v) + ad.eps), lo), hi);
    }
    const float m_hat = m / bc1, v_hat = v / bc2;
    const float qn = q - ad.lr * m_hat / (sqrtf(v_hat) + ad.eps);
    return fminf(fmaxf(qn, lo), hi);
}

