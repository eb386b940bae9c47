/*
 * ppo_oracle.c  --  CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See ppo_oracle.h for the
 * parity status (pinned to the reference graph as executed by oracle/graph_interp.py; no TensorFlow-produced vector exists), conventions and citations.
 *
 * "G:" line numbers refer to the reference's TF MetaGraphDef text proto
 *   /root/reference/resources/ppo_cl/graphs/ppo_cpp_[4_5]_lr_0.0004_cr_0.1610_ent_0.0007.meta.txt
 * other citations are relative to /root/reference/.
 */
#include "ppo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* flat layout                                                                                  */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    int w_pi[ORC_MAX_LAYERS], b_pi[ORC_MAX_LAYERS], w_vf[ORC_MAX_LAYERS], b_vf[ORC_MAX_LAYERS];
    int w_v, b_v, w_mu, b_mu, logstd;
    int total;
} orc_offsets;

static orc_offsets layout(const orc_cfg* c) {
    orc_offsets o;
    int p = 0, in = c->obs_dim;
    for (int l = 0; l < c->n_hidden; ++l) {
        int h = c->hidden[l];
        o.w_pi[l] = p; p += in * h;
        o.b_pi[l] = p; p += h;
        o.w_vf[l] = p; p += in * h;
        o.b_vf[l] = p; p += h;
        in = h;
    }
    o.w_v = p;    p += in;
    o.b_v = p;    p += 1;
    o.w_mu = p;   p += in * c->act_dim;
    o.b_mu = p;   p += c->act_dim;
    o.logstd = p; p += c->act_dim;
    o.total = p;
    return o;
}

int orc_num_tensors(const orc_cfg* c) { return 4 * c->n_hidden + 5; }
int orc_num_params(const orc_cfg* c) { return layout(c).total; }

const char* orc_tensor_info(const orc_cfg* c, int i, int* offset, int* rows, int* cols) {
    static const char* hidden_names[ORC_MAX_LAYERS][4] = {
        {"pi_fc0/w", "pi_fc0/b", "vf_fc0/w", "vf_fc0/b"}, {"pi_fc1/w", "pi_fc1/b", "vf_fc1/w", "vf_fc1/b"},
        {"pi_fc2/w", "pi_fc2/b", "vf_fc2/w", "vf_fc2/b"}, {"pi_fc3/w", "pi_fc3/b", "vf_fc3/w", "vf_fc3/b"},
        {"pi_fc4/w", "pi_fc4/b", "vf_fc4/w", "vf_fc4/b"}, {"pi_fc5/w", "pi_fc5/b", "vf_fc5/w", "vf_fc5/b"},
        {"pi_fc6/w", "pi_fc6/b", "vf_fc6/w", "vf_fc6/b"}, {"pi_fc7/w", "pi_fc7/b", "vf_fc7/w", "vf_fc7/b"}};
    orc_offsets o = layout(c);
    int L = c->n_hidden;
    if (i < 4 * L) {
        int l = i / 4, k = i % 4;
        int in = l == 0 ? c->obs_dim : c->hidden[l - 1], h = c->hidden[l];
        int offs[4] = {o.w_pi[l], o.b_pi[l], o.w_vf[l], o.b_vf[l]};
        *offset = offs[k];
        if (k % 2 == 0) { *rows = in; *cols = h; } else { *rows = h; *cols = 0; }
        return hidden_names[l][k];
    }
    int hl = L ? c->hidden[L - 1] : c->obs_dim;
    switch (i - 4 * L) {
        case 0: *offset = o.w_v;    *rows = hl; *cols = 1;          return "vf/w";
        case 1: *offset = o.b_v;    *rows = 1;  *cols = 0;          return "vf/b";
        case 2: *offset = o.w_mu;   *rows = hl; *cols = c->act_dim; return "pi/w";
        case 3: *offset = o.b_mu;   *rows = c->act_dim; *cols = 0;  return "pi/b";
        case 4: *offset = o.logstd; *rows = 1;  *cols = c->act_dim; return "pi/logstd";
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* dense helpers (double accumulation, one rounding)                                            */
/* ------------------------------------------------------------------------------------------- */
/* y[n,out] = x[n,in] * W[in,out] + b[out]   (MatMul then Add: G:2609-2648) */
static void affine(const float* x, int n, int in, const float* W, const float* b, int out, float* y) {
    double* acc = (double*)malloc(sizeof(double) * (size_t)out);
    for (int r = 0; r < n; ++r) {
        for (int j = 0; j < out; ++j) acc[j] = 0.0;
        const float* xr = x + (size_t)r * in;
        for (int k = 0; k < in; ++k) {
            const double xv = (double)xr[k];
            const float* wr = W + (size_t)k * out;
            for (int j = 0; j < out; ++j) acc[j] += xv * (double)wr[j];
        }
        float* yr = y + (size_t)r * out;
        for (int j = 0; j < out; ++j) yr[j] = (float)acc[j] + b[j];
    }
    free(acc);
}

/* dX[n,in] = dY[n,out] * W^T   (…/MatMul_grad/MatMul) */
static void matmul_nt(const float* dY, int n, int out, const float* W, int in, float* dX) {
    for (int r = 0; r < n; ++r) {
        const float* dyr = dY + (size_t)r * out;
        for (int k = 0; k < in; ++k) {
            const float* wr = W + (size_t)k * out;
            double acc = 0.0;
            for (int j = 0; j < out; ++j) acc += (double)dyr[j] * (double)wr[j];
            dX[(size_t)r * in + k] = (float)acc;
        }
    }
}

/* dW[in,out] = X^T[in,n] * dY[n,out]  (…/MatMul_grad/MatMul_1) ; db[out] = sum_n dY (…/Add_grad/Sum_1) */
static void weight_grads(const float* X, const float* dY, int n, int in, int out, float* dW, float* db) {
    double* acc = (double*)calloc((size_t)in * out + out, sizeof(double));
    double* accb = acc + (size_t)in * out;
    for (int r = 0; r < n; ++r) {
        const float* xr = X + (size_t)r * in;
        const float* dyr = dY + (size_t)r * out;
        for (int k = 0; k < in; ++k) {
            const double xv = (double)xr[k];
            double* a = acc + (size_t)k * out;
            for (int j = 0; j < out; ++j) a[j] += xv * (double)dyr[j];
        }
        for (int j = 0; j < out; ++j) accb[j] += (double)dyr[j];
    }
    for (size_t i = 0; i < (size_t)in * out; ++i) dW[i] = (float)acc[i];
    for (int j = 0; j < out; ++j) db[j] = (float)accb[j];
    free(acc);
}

/* ------------------------------------------------------------------------------------------- */
/* forward with saved activations                                                               */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    float* h_pi[ORC_MAX_LAYERS + 1]; /* h_pi[0] = obs (not owned) ; h_pi[l+1] = tanh(...) [n,h_l] */
    float* h_vf[ORC_MAX_LAYERS + 1];
    float* mu;                        /* [n,A] */
    float* v;                         /* [n]   */
} orc_acts;

static void acts_alloc(const orc_cfg* c, int n, orc_acts* a) {
    for (int l = 0; l < c->n_hidden; ++l) {
        a->h_pi[l + 1] = (float*)malloc(sizeof(float) * (size_t)n * c->hidden[l]);
        a->h_vf[l + 1] = (float*)malloc(sizeof(float) * (size_t)n * c->hidden[l]);
    }
    a->mu = (float*)malloc(sizeof(float) * (size_t)n * c->act_dim);
    a->v = (float*)malloc(sizeof(float) * (size_t)n);
}
static void acts_free(const orc_cfg* c, orc_acts* a) {
    for (int l = 0; l < c->n_hidden; ++l) { free(a->h_pi[l + 1]); free(a->h_vf[l + 1]); }
    free(a->mu); free(a->v);
}

/* Two SEPARATE towers on the same input (G:2609-2675 pi_fc*, G:3061-3127 vf_fc*). The graph's
 * "x = obs + 0.0" (G:1927) is an exact identity in fp32 for finite obs and is not re-stated. */
static void forward_acts(const orc_cfg* c, const float* theta, const float* obs, int n, orc_acts* a) {
    orc_offsets o = layout(c);
    a->h_pi[0] = (float*)obs;
    a->h_vf[0] = (float*)obs;
    int in = c->obs_dim;
    for (int l = 0; l < c->n_hidden; ++l) {
        int h = c->hidden[l];
        affine(a->h_pi[l], n, in, theta + o.w_pi[l], theta + o.b_pi[l], h, a->h_pi[l + 1]);
        affine(a->h_vf[l], n, in, theta + o.w_vf[l], theta + o.b_vf[l], h, a->h_vf[l + 1]);
        for (size_t i = 0; i < (size_t)n * h; ++i) {
            a->h_pi[l + 1][i] = tanhf(a->h_pi[l + 1][i]);
            a->h_vf[l + 1][i] = tanhf(a->h_vf[l + 1][i]);
        }
        in = h;
    }
    affine(a->h_pi[c->n_hidden], n, in, theta + o.w_mu, theta + o.b_mu, c->act_dim, a->mu);  /* G:4843-4882 */
    affine(a->h_vf[c->n_hidden], n, in, theta + o.w_v, theta + o.b_v, 1, a->v);              /* G:4417-4456 */
}

void orc_forward(const orc_cfg* c, const float* theta, const float* obs, int n, float* mu, float* v) {
    orc_acts a;
    acts_alloc(c, n, &a);
    forward_acts(c, theta, obs, n, &a);
    if (mu) memcpy(mu, a.mu, sizeof(float) * (size_t)n * c->act_dim);
    if (v) memcpy(v, a.v, sizeof(float) * (size_t)n);
    acts_free(c, &a);
}

#define ORC_HALF_LOG_2PI 0.9189385175704956f /* G:6531 */
#define ORC_HALF_LOG_2PIE 1.4189385175704956f /* G:10021-10180 */

/* neglogp(a | mu, logstd) (G:6103-6672 / G:9428-9997), z_out optional [A] */
static float neglogp_row(const float* a, const float* mu, const float* logstd_var, int A, float* z_out) {
    double ssq = 0.0, slog = 0.0;
    for (int j = 0; j < A; ++j) {
        const float logstd = mu[j] * 0.0f + logstd_var[j]; /* G:5128-5155 */
        const float sigma = expf(logstd);                  /* G:5779 */
        const float z = (a[j] - mu[j]) / sigma;
        if (z_out) z_out[j] = z;
        ssq += (double)(z * z);
        slog += (double)logstd;
    }
    return 0.5f * (float)ssq + ORC_HALF_LOG_2PI * (float)A + (float)slog;
}

void orc_step(const orc_cfg* c, const float* theta, const float* obs, int n, const float* noise,
              float* action, float* value, float* neglogp) {
    orc_offsets o = layout(c);
    const int A = c->act_dim;
    float* mu = (float*)malloc(sizeof(float) * (size_t)n * A);
    orc_forward(c, theta, obs, n, mu, value);
    for (int r = 0; r < n; ++r) {
        for (int j = 0; j < A; ++j) {
            const float logstd = mu[r * A + j] * 0.0f + theta[o.logstd + j];
            action[r * A + j] = mu[r * A + j] + expf(logstd) * noise[r * A + j]; /* G:5992-6019 */
        }
        neglogp[r] = neglogp_row(action + r * A, mu + r * A, theta + o.logstd, A, 0);
    }
    free(mu);
}

/* ------------------------------------------------------------------------------------------- */
/* loss + backward                                                                              */
/* ------------------------------------------------------------------------------------------- */
static inline float fminf_(float a, float b) { return a < b ? a : b; } /* TF Minimum */
static inline float fmaxf_(float a, float b) { return a > b ? a : b; } /* TF Maximum */

void orc_loss_grad(const orc_cfg* c, const float* theta, const float* obs, const float* actions,
                   const float* advs, const float* returns, const float* old_neglogp,
                   const float* old_values, int n, float cr, float losses[5], float* grad) {
    orc_offsets o = layout(c);
    const int A = c->act_dim, L = c->n_hidden;
    orc_acts a;
    acts_alloc(c, n, &a);
    forward_acts(c, theta, obs, n, &a);

    float* d_mu = (float*)malloc(sizeof(float) * (size_t)n * A);
    float* d_v = (float*)malloc(sizeof(float) * (size_t)n);
    float* z = (float*)malloc(sizeof(float) * (size_t)A);
    double* d_logstd_acc = (double*)calloc((size_t)A, sizeof(double));
    const float g = 1.0f / (float)n;                          /* grad of Mean: 1/N */
    const float lo = 1.0f - cr, hi = 1.0f + cr;               /* G:10470-10918 */
    const float gv = c->vf_coef * 0.5f * g;
    double s_pg = 0, s_vf = 0, s_ent = 0, s_kl = 0, s_cf = 0;

    for (int r = 0; r < n; ++r) {
        const float* mu = a.mu + (size_t)r * A;
        const float nlp = neglogp_row(actions + (size_t)r * A, mu, theta + o.logstd, A, z);
        /* entropy row (G:10021-10180) */
        double erow = 0.0;
        for (int j = 0; j < A; ++j) erow += (double)((mu[j] * 0.0f + theta[o.logstd + j]) + ORC_HALF_LOG_2PIE);
        s_ent += (double)(float)erow;
        /* value loss (G:10213-10837) */
        const float v = a.v[r], R = returns[r], vo = old_values[r];
        const float dvo = v - vo;
        const float vmin = fminf_(dvo, cr);
        const float vclip = vo + fmaxf_(vmin, -cr);
        const float e1 = v - R, e2 = vclip - R;
        const float s1 = e1 * e1, s2 = e2 * e2;
        s_vf += (double)fmaxf_(s1, s2);
        /* policy loss (G:10423-10918) */
        const float ratio = expf(old_neglogp[r] - nlp);
        const float rmin = fminf_(ratio, hi);
        const float rclip = fmaxf_(rmin, lo);
        const float m1 = -advs[r] * ratio, m2 = -advs[r] * rclip;
        s_pg += (double)fmaxf_(m1, m2);
        const float dk = nlp - old_neglogp[r];
        s_kl += (double)(dk * dk);                             /* G:10951-11097 */
        s_cf += (fabsf(ratio - 1.0f) > cr) ? 1.0 : 0.0;        /* G:11118-11290 */

        /* ---- backward heads (TF tie rules: Maximum -> first arg iff x >= y, G:12609,14975) */
        const float sel = (m1 >= m2) ? 1.0f : 0.0f;
        float d_ratio = (-advs[r]) * g * sel;
        /* clip_by_value = Maximum(Minimum(x,hi),lo): passes iff min_out >= lo (G:15357) and x <= hi (G:16113) */
        const float pass = ((rmin >= lo) ? 1.0f : 0.0f) * ((ratio <= hi) ? 1.0f : 0.0f);
        d_ratio += (-advs[r]) * g * (1.0f - sel) * pass;
        const float d_nlp = -(d_ratio * ratio);                /* Exp_grad G:16851, sub */
        for (int j = 0; j < A; ++j) {
            const float sigma = expf(mu[j] * 0.0f + theta[o.logstd + j]);
            /* d mu = d_nlp * (-(z/sigma)) ; plus the exact-zero term from (mu*0) (AddN_3 G:22656) */
            const float dl = d_nlp * (1.0f - z[j] * z[j]) - c->ent_coef * g;      /* AddN_2 G:21299 */
            d_mu[(size_t)r * A + j] = d_nlp * (-(z[j] / sigma)) + dl * 0.0f;
            d_logstd_acc[j] += (double)dl;
        }
        const float selv = (s1 >= s2) ? 1.0f : 0.0f;          /* G:14975 */
        const float passv = ((vmin >= -cr) ? 1.0f : 0.0f) * ((dvo <= cr) ? 1.0f : 0.0f); /* G:17477,18071 */
        d_v[r] = gv * selv * (2.0f * e1) + gv * (1.0f - selv) * (2.0f * e2) * passv;     /* AddN_1 G:19571 */
    }
    losses[0] = (float)s_pg / (float)n;
    losses[1] = 0.5f * ((float)s_vf / (float)n);
    losses[2] = (float)s_ent / (float)n;
    losses[3] = 0.5f * ((float)s_kl / (float)n);
    losses[4] = (float)s_cf / (float)n;

    if (grad) {
        memset(grad, 0, sizeof(float) * (size_t)o.total);
        for (int j = 0; j < A; ++j) grad[o.logstd + j] = (float)d_logstd_acc[j];
        const int hl = L ? c->hidden[L - 1] : c->obs_dim;
        /* heads */
        weight_grads(a.h_pi[L], d_mu, n, hl, A, grad + o.w_mu, grad + o.b_mu);
        weight_grads(a.h_vf[L], d_v, n, hl, 1, grad + o.w_v, grad + o.b_v);
        if (L > 0) {
            float* dh_pi = (float*)malloc(sizeof(float) * (size_t)n * hl);
            float* dh_vf = (float*)malloc(sizeof(float) * (size_t)n * hl);
            matmul_nt(d_mu, n, A, theta + o.w_mu, hl, dh_pi);
            matmul_nt(d_v, n, 1, theta + o.w_v, hl, dh_vf);
            for (int l = L - 1; l >= 0; --l) {
                const int h = c->hidden[l];
                const int in = l == 0 ? c->obs_dim : c->hidden[l - 1];
                /* TanhGrad: dy * (1 - y*y) (G:21272,22131,23078,23408) */
                for (size_t i = 0; i < (size_t)n * h; ++i) {
                    const float yp = a.h_pi[l + 1][i], yv = a.h_vf[l + 1][i];
                    dh_pi[i] = dh_pi[i] * (1.0f - yp * yp);
                    dh_vf[i] = dh_vf[i] * (1.0f - yv * yv);
                }
                weight_grads(a.h_pi[l], dh_pi, n, in, h, grad + o.w_pi[l], grad + o.b_pi[l]);
                weight_grads(a.h_vf[l], dh_vf, n, in, h, grad + o.w_vf[l], grad + o.b_vf[l]);
                if (l > 0) { /* first-layer dX is never needed (obs is a placeholder) */
                    float* nx_pi = (float*)malloc(sizeof(float) * (size_t)n * in);
                    float* nx_vf = (float*)malloc(sizeof(float) * (size_t)n * in);
                    matmul_nt(dh_pi, n, h, theta + o.w_pi[l], in, nx_pi);
                    matmul_nt(dh_vf, n, h, theta + o.w_vf[l], in, nx_vf);
                    free(dh_pi); free(dh_vf);
                    dh_pi = nx_pi; dh_vf = nx_vf;
                }
            }
            free(dh_pi); free(dh_vf);
        }
    }
    free(d_mu); free(d_v); free(z); free(d_logstd_acc);
    acts_free(c, &a);
}

float orc_clip_by_global_norm(const orc_cfg* c, float* grad) {
    const int nt = orc_num_tensors(c);
    double stack = 0.0; /* sum over the 13 L2Loss values in TF order (G:23738-24269) */
    for (int i = 0; i < nt; ++i) {
        int off, rows, cols;
        orc_tensor_info(c, i, &off, &rows, &cols);
        const int cnt = rows * (cols ? cols : 1);
        double s = 0.0;
        for (int k = 0; k < cnt; ++k) s += (double)grad[off + k] * (double)grad[off + k];
        stack += (double)(float)(s / 2.0); /* L2Loss = sum(x^2)/2, one fp32 value per tensor */
    }
    const float norm = sqrtf((float)stack * 2.0f);                       /* G:24218-24269 */
    const float cn = c->max_grad_norm;
    /* scale = clip_norm * min(1/norm, 1/clip_norm) (G:24289-24472) ; NaN if norm not finite (G:24493-24543) */
    float scale = cn * fminf_(1.0f / norm, 1.0f / cn);
    if (!isfinite(norm)) scale = NAN;
    const int P = orc_num_params(c);
    for (int k = 0; k < P; ++k) grad[k] = grad[k] * scale;
    return norm;
}

void orc_adam(const orc_cfg* c, float* theta, float* m, float* v, const float* grad, float lr, float pw[2]) {
    const int P = orc_num_params(c);
    const float b1 = c->adam_beta1, b2 = c->adam_beta2, eps = c->adam_eps;
    const float alpha = lr * sqrtf(1.0f - pw[1]) / (1.0f - pw[0]);     /* TF-1.14 ApplyAdam functor */
    for (int k = 0; k < P; ++k) {
        const float g = grad[k];
        m[k] = m[k] + (g - m[k]) * (1.0f - b1);
        v[k] = v[k] + (g * g - v[k]) * (1.0f - b2);
        theta[k] = theta[k] - (m[k] * alpha) / (sqrtf(v[k]) + eps);
    }
    pw[0] = pw[0] * b1;                                                 /* G:31217-31342: after the applies */
    pw[1] = pw[1] * b2;
}

float orc_train_step(const orc_cfg* c, float* theta, float* m, float* v, float pw[2], float lr,
                     float cliprange, const float* obs, const float* actions, const float* advs,
                     const float* returns, const float* old_neglogp, const float* old_values, int n,
                     float losses[5], float* grad_scratch) {
    orc_loss_grad(c, theta, obs, actions, advs, returns, old_neglogp, old_values, n, cliprange, losses,
                  grad_scratch);
    const float norm = orc_clip_by_global_norm(c, grad_scratch);
    orc_adam(c, theta, m, v, grad_scratch, lr, pw);
    return norm;
}

/* ------------------------------------------------------------------------------------------- */
/* host-side numerics                                                                           */
/* ------------------------------------------------------------------------------------------- */
void orc_adv_normalize(const float* returns, const float* values, int n, float* advs) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) { advs[i] = returns[i] - values[i]; s += (double)advs[i]; }
    const float mean = (float)(s / (double)n);                         /* ppo2.hpp:403 */
    double sq = 0.0;
    for (int i = 0; i < n; ++i) { advs[i] = advs[i] - mean; sq += (double)(advs[i] * advs[i]); }
    const float var = (float)sq / (float)n;                            /* ppo2.hpp:405 */
    const float denom = (float)((double)sqrtf(var) + 1e-8);            /* ppo2.hpp:406 */
    for (int i = 0; i < n; ++i) advs[i] = advs[i] / denom;
}

void orc_gae(const float* rewards, const float* values, const float* dones, const float* last_values,
             const float* last_dones, int T, int E, float gamma, float lam, float* returns) {
    for (int e = 0; e < E; ++e) {
        float last = 0.0f;
        for (int t = T - 1; t >= 0; --t) {
            float nnt, nv;
            if (t == T - 1) { nnt = 1.0f - last_dones[e]; nv = last_values[e]; }          /* runner.hpp:177-180 */
            else { nnt = 1.0f - dones[(size_t)(t + 1) * E + e]; nv = values[(size_t)(t + 1) * E + e]; }
            const float delta = rewards[(size_t)t * E + e] + gamma * (nv * nnt) - values[(size_t)t * E + e];
            last = delta + (gamma * lam) * (nnt * last);                                   /* runner.hpp:187-188 */
            returns[(size_t)t * E + e] = last + values[(size_t)t * E + e];                 /* runner.hpp:190 */
        }
    }
}

void orc_rstats_init(orc_rstats* s, int dim, float* mean_buf, float* var_buf) {
    s->dim = dim; s->count = 1e-6; s->mean = mean_buf; s->var = var_buf; /* running_statistics.hpp:17-21 */
    for (int j = 0; j < dim; ++j) { mean_buf[j] = 0.0f; var_buf[j] = 1.0f; }
}

void orc_rstats_update(orc_rstats* s, const float* batch, int rows) {
    const int D = s->dim;
    const double nb = (double)rows;
    const double tot = s->count + nb;
    for (int j = 0; j < D; ++j) {
        double sum = 0.0;
        for (int r = 0; r < rows; ++r) sum += (double)batch[(size_t)r * D + j];
        const float bmean = (float)(sum / nb);                              /* colwise().mean() :38-39 */
        double m2 = 0.0;
        for (int r = 0; r < rows; ++r) { const float d = batch[(size_t)r * D + j] - bmean; m2 += (double)(d * d); }
        const float bvar = (float)m2 / (float)nb;                            /* :51-54 population variance */
        const float delta = bmean - s->mean[j];                              /* :90 */
        const float new_mean = s->mean[j] + (delta * (float)nb) / (float)tot; /* :94 */
        const float m_a = s->var[j] * (float)s->count;                       /* :97 */
        const float m_b = bvar * (float)nb;                                  /* :98 */
        const float M2 = m_a + m_b + (((delta * delta) * (float)s->count) * (float)nb) / (float)tot; /* :100 */
        s->mean[j] = new_mean;
        s->var[j] = M2 / (float)tot;                                         /* :101 */
    }
    s->count = nb + s->count;                                                /* :103 */
}

void orc_normalize_obs(orc_rstats* s, const float* obs, int rows, int training, float clip, float eps,
                       float* out) {
    const int D = s->dim;
    if (training) orc_rstats_update(s, obs, rows);                           /* env_normalize.hpp:96-97 */
    for (int j = 0; j < D; ++j) {
        const float inv = 1.0f / sqrtf(s->var[j] + eps);                     /* cwiseSqrt().cwiseInverse() :100 */
        for (int r = 0; r < rows; ++r) {
            float y = (obs[(size_t)r * D + j] - s->mean[j]) * inv;
            y = y > -clip ? y : -clip;                                       /* matrix_clamp.hpp:33 cwiseMax(lo) */
            y = y < clip ? y : clip;                                         /* cwiseMin(hi) */
            out[(size_t)r * D + j] = y;
        }
    }
}

void orc_normalize_reward(orc_rstats* s, float* ret, const float* rew, const float* dones, int rows,
                          int training, float gamma, float clip, float eps, float* out) {
    for (int r = 0; r < rows; ++r) ret[r] = ret[r] * gamma + rew[r];         /* env_normalize.hpp:71 */
    if (training) orc_rstats_update(s, ret, rows);                           /* :76-77 */
    const float inv = 1.0f / sqrtf(s->var[0] + eps);                         /* :80 */
    for (int r = 0; r < rows; ++r) {
        float y = rew[r] * inv;
        y = y > -clip ? y : -clip;
        y = y < clip ? y : clip;                                             /* :82 */
        out[r] = y;
        ret[r] = ret[r] * (1.0f - dones[r]);                                 /* :88 */
    }
}

/* ------------------------------------------------------------------------------------------- */
/* seeded synthetic env                                                                         */
/* ------------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

uint32_t orc_hash(uint32_t seed, uint32_t env, uint32_t step, uint32_t lane) {
    const uint64_t a = ((uint64_t)seed << 32) | (uint64_t)env;
    const uint64_t b = ((uint64_t)step << 32) | (uint64_t)lane;
    return (uint32_t)(splitmix64(splitmix64(a) ^ b) >> 32);
}

static inline float u32_to_sym_unit(uint32_t h) { return (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; }

void orc_seeded_env_step(uint32_t seed, int env0, int n_envs, uint32_t step, int obs_dim, float* obs,
                         float* rew, float* dones) {
    for (int e = 0; e < n_envs; ++e) {
        const uint32_t ge = (uint32_t)(env0 + e);
        for (int j = 0; j < obs_dim; ++j) obs[(size_t)e * obs_dim + j] = u32_to_sym_unit(orc_hash(seed, ge, step, (uint32_t)j));
        if (rew) rew[e] = u32_to_sym_unit(orc_hash(seed, ge, step, (uint32_t)obs_dim));
        if (dones) dones[e] = (orc_hash(seed, ge, step, (uint32_t)obs_dim + 1u) % 300u == 0u) ? 1.0f : 0.0f;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* update loop (ppo2.hpp:264-335)                                                               */
/* ------------------------------------------------------------------------------------------- */
void orc_update(const orc_cfg* c, float* theta, float* m, float* v, float pw[2], const orc_rollout* ro,
                const int32_t* perms, int epochs, int nminibatches, float lr, float cliprange,
                float* loss_rows, float mean_losses[5]) {
    const int E = ro->E, T = ro->T, B = E * T, M = B / nminibatches;
    const int O = c->obs_dim, A = c->act_dim;
    float* obs = (float*)malloc(sizeof(float) * (size_t)M * O);
    float* act = (float*)malloc(sizeof(float) * (size_t)M * A);
    float* ret = (float*)malloc(sizeof(float) * (size_t)M);
    float* val = (float*)malloc(sizeof(float) * (size_t)M);
    float* nlp = (float*)malloc(sizeof(float) * (size_t)M);
    float* adv = (float*)malloc(sizeof(float) * (size_t)M);
    float* grad = (float*)malloc(sizeof(float) * (size_t)orc_num_params(c));
    int32_t* inv = (int32_t*)malloc(sizeof(int32_t) * (size_t)B);
    double acc[5] = {0, 0, 0, 0, 0};
    for (int ep = 0; ep < epochs; ++ep) {
        const int32_t* perm = perms + (size_t)ep * B;
        for (int i = 0; i < B; ++i) inv[perm[i]] = i;          /* out.row(perm[i]) = in.row(i) (ppo2.hpp:291-296) */
        for (int k = 0; k < nminibatches; ++k) {
            for (int i = 0; i < M; ++i) {
                const int r = inv[k * M + i];                    /* flattened env-major row (runner.hpp:136-152) */
                const int s = orc_row_to_storage(r, T, E);
                memcpy(obs + (size_t)i * O, ro->obs + (size_t)s * O, sizeof(float) * O);
                memcpy(act + (size_t)i * A, ro->actions + (size_t)s * A, sizeof(float) * A);
                ret[i] = ro->returns[s]; val[i] = ro->values[s]; nlp[i] = ro->neglogp[s];
            }
            orc_adv_normalize(ret, val, M, adv);
            float* L5 = loss_rows + (size_t)(ep * nminibatches + k) * 5;
            orc_train_step(c, theta, m, v, pw, lr, cliprange, obs, act, adv, ret, nlp, val, M, L5, grad);
            for (int j = 0; j < 5; ++j) acc[j] += (double)L5[j];
        }
    }
    for (int j = 0; j < 5; ++j) mean_losses[j] = (float)(acc[j] / (double)(epochs * nminibatches));
    free(obs); free(act); free(ret); free(val); free(nlp); free(adv); free(grad); free(inv);
}
