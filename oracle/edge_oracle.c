/*
 * edge_oracle.c - plain-C CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built
 * from this file (oracle/_build/libedge_oracle.so); the product never links or calls it.
 *
 * What it restates (reference file:line, all under /root/reference):
 *   eo_reset / eo_step     EdgeSimGym.reset / step / _generate_tasks_and_obs / _get_obs /
 *                          _calculate_reward / _maybe_apply_env_exploration   MRRL.PY:188-521
 *   eo_metrics             episode metrics                                     MRRL.PY:470-504
 *   eo_gae                 MAPPO.compute_gae (fp64, sequential)                mappo_algorithm.py:241-262
 *   np_pairwise_sum        numpy's float add.reduce summation order, which np.mean/np.sum in
 *                          MRRL.PY:334-336,478,483 inherit (numpy/_core/src/umath/loops_utils.h.src)
 *   np_percentile_linear   np.percentile(x, 99) default method            numpy/lib/_function_base_impl.py
 *
 * Two draw sources: a structured tape recorded from the reference (parity with the reference's
 * MT19937 stream) and the build-defined Philox4x32-10 counter stream (see oracle/philox_ref.py for
 * the slow Python form this file is pinned against).  Everything is float64 in the reference's
 * evaluation order; compile with -ffp-contract=off so no multiply-add is fused.
 *
 * Pinned by tests/test_oracle_env.py against the golden fixtures made from the imported reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EO_MAX_N 64

typedef struct {
    int32_t n;              /* agents == servers (MRRL.PY:128,150) */
    int32_t horizon;        /* episode_length */
    int32_t rate_mode;      /* 0: midpoint, 1: cyclic, 2: random (MRRL.PY:225-235) */
    int32_t explore_off;    /* disable_env_exploration */
    double cap[EO_MAX_N];
    double rate_lo, rate_hi, cycle;
    double wl_lo, wl_hi, data_lo, data_hi, ddl_lo, ddl_hi;
    double bw_lo, bw_hi, lat_lo, lat_hi;
    double w_lat, w_en, w_viol;
    double dt_min, dt_max, window0;
    double eps0, eps1;
    int64_t decay_steps;
} eo_config;

typedef struct {
    const double *explore_u, *explore_res, *bw_u, *lat;   /* [B][N], indexed by acting slot */
    const int32_t *explore_tgt;                            /* [B][N] */
    const double *rate_u, *dt_raw;                         /* [B] */
    const int32_t *n_arr;                                  /* [B] */
    const double *task_wl, *task_data, *task_ddl_off;      /* [B][A] */
    const int32_t *task_decider;                           /* [B][A] */
    int32_t a_slots;
} eo_tape;

typedef struct {
    double load[EO_MAX_N], queue[EO_MAX_N];
    double t_wl[EO_MAX_N], t_data[EO_MAX_N], t_ddl[EO_MAX_N];
    int32_t rank[EO_MAX_N];   /* -1 = no pending task */
    int32_t n_pending;
    int32_t step;
    double sim_time, elapsed;
    /* episode bookkeeping (MRRL.PY:182-186) */
    double *lat, *en;         /* growing lists */
    int64_t n_lat, cap_lat;
    int64_t n_viol, n_completed;
    double usage_sum[EO_MAX_N];
    int64_t n_usage;
    uint32_t episode;
    int64_t n_draws;          /* draws consumed in the last call */
} eo_env;

typedef struct {
    eo_config cfg;
    int64_t B;
    eo_env *env;
    uint64_t seed;
    uint64_t env_offset;
    int64_t explore_counter;  /* the reference's class-level _global_step_counter (MRRL.PY:122) */
} eo_handle;

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10 and the deterministic math of the throughput mode                             */
/* ------------------------------------------------------------------------------------------ */
enum { P_DT = 0, P_POIS, P_TASK_A, P_TASK_B, P_EXPL, P_EXPL_RES, P_NET };

typedef struct { uint32_t x[4]; } blk_t;

static blk_t philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    blk_t b = {{c0, c1, c2, c3}};
    return b;
}

static double u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

static double det_log(double x)
{
    int e;
    double m = frexp(x, &e);
    if (m < 0.7071067811865476) { m = m * 2.0; e -= 1; }
    double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 25.0;
    for (int den = 23; den >= 3; den -= 2) p = p * z + 1.0 / (double)den;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}

static double inv_fact(int i)
{
    double f = 1.0;
    for (int k = 2; k <= i; ++k) f *= (double)k;
    return 1.0 / f;
}

static double det_exp(double x)
{
    double k = floor(x * 1.4426950408889634 + 0.5);
    double r = x - k * 0.6931471805599453;
    double p = inv_fact(14);
    for (int i = 13; i >= 0; --i) p = p * r + inv_fact(i);
    return ldexp(p, (int)k);
}

static double det_sin2pi(double phase)
{
    double j = floor(phase * 4.0 + 0.5);
    double x = 6.283185307179586 * (phase - j * 0.25);
    double z = x * x, v;
    int q = (int)j & 3;
    if (q & 1) {
        double p = inv_fact(16), sgn = -1.0;
        for (int i = 14; i >= 0; i -= 2) { p = p * z + sgn * inv_fact(i); sgn = -sgn; }
        v = p;
    } else {
        double p = -inv_fact(15), sgn = 1.0;
        for (int i = 13; i >= 1; i -= 2) { p = p * z + sgn * inv_fact(i); sgn = -sgn; }
        v = p * x;
    }
    return (q & 2) ? -v : v;
}

/* ------------------------------------------------------------------------------------------ */
/* numpy summation order and percentile                                                       */
/* ------------------------------------------------------------------------------------------ */
double np_pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double r = -0.0;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        int64_t i;
        for (i = 0; i < 8; ++i) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

static double np_sum(const double *a, int64_t n) { return 0.0 + np_pairwise_sum(a, n); }

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

double np_percentile_linear(const double *x, int64_t n, double q)
{
    double *s = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(s, x, sizeof(double) * (size_t)n);
    qsort(s, (size_t)n, sizeof(double), cmp_double);
    double vi = (double)(n - 1) * (q / 100.0);
    double fl = floor(vi);
    int64_t lo = (int64_t)fl, hi = lo + 1;
    if (vi >= (double)(n - 1)) { lo = n - 1; hi = n - 1; }
    if (lo < 0) lo = 0;
    if (hi > n - 1) hi = n - 1;
    double a = s[lo], b = s[hi], t = vi - fl, d = b - a;
    double out = (t >= 0.5) ? b - d * (1.0 - t) : a + d * t;
    free(s);
    return out;
}

/* ------------------------------------------------------------------------------------------ */
/* handle                                                                                     */
/* ------------------------------------------------------------------------------------------ */
eo_handle *eo_create(const eo_config *cfg, int64_t B, uint64_t seed, uint64_t env_offset)
{
    if (cfg->n < 1 || cfg->n > EO_MAX_N || B < 1) return NULL;
    eo_handle *h = (eo_handle *)calloc(1, sizeof(eo_handle));
    h->cfg = *cfg;
    h->B = B;
    h->seed = seed;
    h->env_offset = env_offset;
    h->env = (eo_env *)calloc((size_t)B, sizeof(eo_env));
    return h;
}

void eo_destroy(eo_handle *h)
{
    if (!h) return;
    for (int64_t b = 0; b < h->B; ++b) { free(h->env[b].lat); free(h->env[b].en); }
    free(h->env);
    free(h);
}

void eo_set_explore(eo_handle *h, int32_t explore_off, int64_t counter)
{
    h->cfg.explore_off = explore_off;
    h->explore_counter = counter;
}
int64_t eo_get_explore_counter(const eo_handle *h) { return h->explore_counter; }
void eo_set_seed(eo_handle *h, uint64_t seed) { h->seed = seed; for (int64_t b = 0; b < h->B; ++b) h->env[b].episode = 0; }

static void push_task(eo_env *e, double lat, double en)
{
    if (e->n_lat == e->cap_lat) {
        e->cap_lat = e->cap_lat ? e->cap_lat * 2 : 1024;
        e->lat = (double *)realloc(e->lat, sizeof(double) * (size_t)e->cap_lat);
        e->en = (double *)realloc(e->en, sizeof(double) * (size_t)e->cap_lat);
    }
    e->lat[e->n_lat] = lat;
    e->en[e->n_lat] = en;
    e->n_lat++;
}

static double arrival_rate(const eo_config *c, double t, int philox_mode, double rate_u)
{
    if (c->rate_lo == c->rate_hi) return c->rate_lo;
    if (c->rate_mode == 1) {
        double cyc = c->cycle > 1e-6 ? c->cycle : 1e-6;
        double phase = fmod(t, cyc) / cyc;
        double s = philox_mode ? det_sin2pi(phase) : sin(2 * 3.141592653589793 * phase);
        return c->rate_lo + (0.5 + 0.5 * s) * (c->rate_hi - c->rate_lo);
    }
    if (c->rate_mode == 2) return rate_u;
    return 0.5 * (c->rate_lo + c->rate_hi);
}

/* MRRL.PY:340-383.  obs rows are written for the first task of each decider, zero elsewhere. */
static double arrivals(eo_handle *h, int64_t b, double t, int advance, const eo_tape *tape, float *obs)
{
    const eo_config *c = &h->cfg;
    eo_env *e = &h->env[b];
    const int n = c->n, w = n + 4;
    const uint32_t envid = (uint32_t)(h->env_offset + (uint64_t)b);
    for (int i = 0; i < n; ++i) e->rank[i] = -1;
    e->n_pending = 0;
    memset(obs, 0, sizeof(float) * (size_t)(n * w));

    double rate_u = 0.0;
    if (c->rate_mode == 2 && c->rate_lo != c->rate_hi) {
        if (tape) rate_u = tape->rate_u[b];
        else { blk_t k = philox(envid, e->episode, (uint32_t)e->step, (P_DT << 24), h->seed);
               rate_u = c->rate_lo + (c->rate_hi - c->rate_lo) * u53(k.x[2], k.x[3]); }
        e->n_draws++;
    }
    double rate = arrival_rate(c, t, tape == NULL, rate_u);
    double window;
    if (advance) {
        double raw;
        double r = rate > 1e-6 ? rate : 1e-6;
        if (tape) raw = tape->dt_raw[b];
        else { blk_t k = philox(envid, e->episode, (uint32_t)e->step, (P_DT << 24), h->seed);
               raw = (1.0 / r) * (-det_log(1.0 - u53(k.x[0], k.x[1]))); }
        e->n_draws++;
        window = raw < c->dt_min ? c->dt_min : (raw > c->dt_max ? c->dt_max : raw);
    } else window = c->window0;

    double lam = (rate > 1e-6 ? rate : 1e-6) * window;
    int64_t n_new;
    if (tape) n_new = tape->n_arr[b];
    else {
        double limit = det_exp(-lam), prod = 1.0;
        n_new = 0;
        for (uint32_t i = 0;; ++i) {
            blk_t k = philox(envid, e->episode, (uint32_t)e->step, (P_POIS << 24) | (i >> 1), h->seed);
            double u = (i & 1) ? u53(k.x[2], k.x[3]) : u53(k.x[0], k.x[1]);
            prod = prod * u;
            if (prod > limit && n_new < 4096) n_new++; else break;
        }
    }
    e->n_draws++;

    double ratio[EO_MAX_N];
    for (int i = 0; i < n; ++i) ratio[i] = e->load[i] / c->cap[i];
    for (int64_t j = 0; j < n_new; ++j) {
        double wl, data, off; int dec;
        if (tape) {
            int64_t o = b * tape->a_slots + j;
            wl = tape->task_wl[o]; data = tape->task_data[o]; off = tape->task_ddl_off[o]; dec = tape->task_decider[o];
        } else {
            blk_t ka = philox(envid, e->episode, (uint32_t)e->step, (P_TASK_A << 24) | (uint32_t)j, h->seed);
            blk_t kb = philox(envid, e->episode, (uint32_t)e->step, (P_TASK_B << 24) | (uint32_t)j, h->seed);
            wl = c->wl_lo + (c->wl_hi - c->wl_lo) * u53(ka.x[0], ka.x[1]);
            data = c->data_lo + (c->data_hi - c->data_lo) * u53(ka.x[2], ka.x[3]);
            off = c->ddl_lo + (c->ddl_hi - c->ddl_lo) * u53(kb.x[0], kb.x[1]);
            dec = (int)(((uint64_t)kb.x[2] * (uint64_t)n) >> 32);
        }
        e->n_draws += 4;
        double deadline = t + off;
        if (e->rank[dec] >= 0) { e->queue[dec] += 1; continue; }
        e->rank[dec] = e->n_pending++;
        e->t_wl[dec] = wl; e->t_data[dec] = data; e->t_ddl[dec] = deadline;
        float *row = obs + dec * w;
        double rem = deadline - t;
        row[0] = (float)ratio[dec];
        row[1] = (float)e->queue[dec];
        row[2] = (float)data;
        row[3] = (float)wl;
        row[4] = (float)(rem > 0 ? rem : 0.0);
        int o = 5;
        for (int i = 0; i < n; ++i) if (i != dec) row[o++] = (float)ratio[i];
    }
    double decay = advance ? window : 0.0;
    for (int i = 0; i < n; ++i) {
        double processed = c->cap[i] * decay;
        e->load[i] -= processed;
        if (!(e->load[i] > 0)) e->load[i] = 0.0;
        if (advance) { e->queue[i] -= 1; if (!(e->queue[i] > 0)) e->queue[i] = 0.0; }
        double den = c->cap[i] + 1e-9;
        e->usage_sum[i] += den > 0 ? processed / den : 0.0;
    }
    e->n_usage++;
    return advance ? window : 0.0;
}

static void export_order(const eo_env *e, int n, uint32_t *mask_lo, int8_t *rank)
{
    uint64_t m = 0;
    for (int i = 0; i < n; ++i) { rank[i] = (int8_t)e->rank[i]; if (e->rank[i] >= 0) m |= 1ull << i; }
    mask_lo[0] = (uint32_t)m; mask_lo[1] = (uint32_t)(m >> 32);
}

/* tape == NULL selects the Philox stream.  new_episode != 0 advances the Philox episode index. */
void eo_reset(eo_handle *h, const eo_tape *tape, float *obs, uint32_t *mask2, int8_t *rank)
{
    const int n = h->cfg.n;
    for (int64_t b = 0; b < h->B; ++b) {
        eo_env *e = &h->env[b];
        memset(e->load, 0, sizeof e->load); memset(e->queue, 0, sizeof e->queue);
        memset(e->usage_sum, 0, sizeof e->usage_sum);
        e->n_usage = 0; e->step = 0; e->sim_time = 0; e->elapsed = 0;
        e->n_lat = 0; e->n_viol = 0; e->n_completed = 0; e->n_draws = 0;
        arrivals(h, b, e->sim_time, 0, tape, obs + b * n * (n + 4));
        export_order(e, n, mask2 + 2 * b, rank + b * n);
    }
}

void eo_next_episode(eo_handle *h) { for (int64_t b = 0; b < h->B; ++b) h->env[b].episode++; }

/* MRRL.PY:386-468.  placement < 0 means "no action supplied for this agent". */
void eo_step(eo_handle *h, const int32_t *placement, const float *resource, const eo_tape *tape,
             float *obs, uint32_t *mask2, int8_t *rank, double *reward, uint8_t *emitted, uint8_t *done,
             double *step_lat /* [B][N] or NULL */, int32_t *n_draws /* [B] or NULL */)
{
    const eo_config *c = &h->cfg;
    const int n = c->n;
    double eps = 0.0;
    if (!c->explore_off) {
        int64_t span = c->decay_steps > 1 ? c->decay_steps : 1;
        double frac = (double)h->explore_counter / (double)span;
        if (frac > 1.0) frac = 1.0;
        eps = c->eps0 - (c->eps0 - c->eps1) * frac;
    }
    for (int64_t b = 0; b < h->B; ++b) {
        eo_env *e = &h->env[b];
        const uint32_t envid = (uint32_t)(h->env_offset + (uint64_t)b);
        e->step += 1;
        e->n_draws = 0;
        const double t = e->sim_time;
        double lats[EO_MAX_N], ens[EO_MAX_N];
        int k = 0, nviol = 0;
        for (int r = 0; r < e->n_pending; ++r) {
            int a = -1;
            for (int i = 0; i < n; ++i) if (e->rank[i] == r) a = i;
            int target = placement[b * n + a];
            if (target < 0) continue;
            float res = resource[b * n + a];
            if (!c->explore_off) {
                double u; blk_t kx;
                if (tape) u = tape->explore_u[b * n + k];
                else { kx = philox(envid, e->episode, (uint32_t)e->step, (P_EXPL << 24) | (uint32_t)a, h->seed); u = u53(kx.x[0], kx.x[1]); }
                e->n_draws++;
                if (!(u > eps)) {
                    double r64;
                    if (tape) { target = tape->explore_tgt[b * n + k]; r64 = tape->explore_res[b * n + k]; }
                    else {
                        target = (int)(((uint64_t)kx.x[2] * (uint64_t)n) >> 32);
                        blk_t k2 = philox(envid, e->episode, (uint32_t)e->step, (P_EXPL_RES << 24) | (uint32_t)a, h->seed);
                        r64 = 0.1 + (1.0 - 0.1) * u53(k2.x[0], k2.x[1]);
                    }
                    e->n_draws += 2;
                    res = (float)r64;
                }
            }
            double t_trans = 0.0;
            if (target != a) {
                double bw_u, lat;
                if (tape) { bw_u = tape->bw_u[b * n + k]; lat = tape->lat[b * n + k]; }
                else {
                    blk_t kn = philox(envid, e->episode, (uint32_t)e->step, (P_NET << 24) | (uint32_t)a, h->seed);
                    bw_u = c->bw_lo + (c->bw_hi - c->bw_lo) * u53(kn.x[0], kn.x[1]);
                    lat = c->lat_lo + (c->lat_hi - c->lat_lo) * u53(kn.x[2], kn.x[3]);
                }
                e->n_draws += 2;
                double bw = bw_u * 1e6 / 8;
                t_trans = (e->t_data[a] / bw) + lat;
            }
            double t_queue = e->load[target] / c->cap[target];
            double alloc = c->cap[target] * (double)res;
            double g = alloc / 1e9;
            double t_exec = e->t_wl[a] / g;
            double energy = e->t_wl[a] * g;
            double t_total = t_trans + t_queue + t_exec;
            lats[k] = t_total; ens[k] = energy;
            if ((t + t_total) > e->t_ddl[a]) nviol++; else e->n_completed++;
            e->load[target] += t_exec * g;
            e->queue[target] += 1;
            if (step_lat) step_lat[b * n + k] = t_total;
            push_task(e, t_total, energy);
            k++;
        }
        if (step_lat) for (int i = k; i < n; ++i) step_lat[b * n + i] = NAN;
        e->n_viol += nviol;

        double dt = arrivals(h, b, t, 1, tape, obs + b * n * (n + 4));
        if (dt > 0) { e->sim_time += dt; e->elapsed += dt; }
        double rew = 0.0;
        if (k > 0) {
            double cost = (c->w_lat * (np_sum(lats, k) / (double)k)) + (c->w_en * (np_sum(ens, k) / (double)k))
                          + (c->w_viol * (double)nviol);
            rew = -cost;
        }
        reward[b] = rew;
        emitted[b] = e->n_pending > 0;
        done[b] = e->step >= c->horizon;
        export_order(e, n, mask2 + 2 * b, rank + b * n);
        if (n_draws) n_draws[b] = (int32_t)e->n_draws;
    }
    if (!c->explore_off) h->explore_counter += 1;
}

/* out6 per env: avg_latency_ms, p99_latency_ms, avg_energy_J, violation_rate, throughput_tps, jain */
void eo_metrics(const eo_handle *h, double *out6)
{
    const int n = h->cfg.n;
    for (int64_t b = 0; b < h->B; ++b) {
        const eo_env *e = &h->env[b];
        double *o = out6 + 6 * b;
        int64_t total = e->n_lat ? e->n_lat : 1;
        if (e->n_lat) {
            double *ms = (double *)malloc(sizeof(double) * (size_t)e->n_lat);
            for (int64_t i = 0; i < e->n_lat; ++i) ms[i] = e->lat[i] * 1000;
            o[0] = np_sum(ms, e->n_lat) / (double)e->n_lat;
            o[1] = np_percentile_linear(ms, e->n_lat, 99.0);
            o[2] = np_sum(e->en, e->n_lat) / (double)e->n_lat;
            free(ms);
        } else { o[0] = 0; o[1] = 0; o[2] = NAN; }
        o[3] = (double)e->n_viol / (double)total;
        o[4] = (double)e->n_completed / (e->elapsed > 1e-6 ? e->elapsed : 1e-6);
        double m[EO_MAX_N], sq[EO_MAX_N];
        for (int i = 0; i < n; ++i) { m[i] = e->n_usage ? e->usage_sum[i] / (double)e->n_usage : 0.0; sq[i] = m[i] * m[i]; }
        double s1 = np_sum(m, n), s2 = np_sum(sq, n);
        o[5] = (s1 * s1) / ((double)n * s2 + 1e-9);
    }
}

/* state export for parity checks */
void eo_get_state(const eo_handle *h, double *load, double *queue, double *sim_time, int32_t *step,
                  int64_t *counts3 /* n_tasks, n_viol, n_completed */, double *elapsed)
{
    const int n = h->cfg.n;
    for (int64_t b = 0; b < h->B; ++b) {
        const eo_env *e = &h->env[b];
        for (int i = 0; i < n; ++i) { load[b * n + i] = e->load[i]; queue[b * n + i] = e->queue[i]; }
        sim_time[b] = e->sim_time; step[b] = e->step; elapsed[b] = e->elapsed;
        counts3[3 * b] = e->n_lat; counts3[3 * b + 1] = e->n_viol; counts3[3 * b + 2] = e->n_completed;
    }
}

/* mappo_algorithm.py:241-262: sequential reverse recursion in float64 over one flattened list. */
void eo_gae(const double *rewards, const double *values, const uint8_t *dones, int64_t L, double next_value,
            double gamma, double lam, double *adv, double *ret)
{
    double gae = 0.0;
    for (int64_t t = L - 1; t >= 0; --t) {
        double nv = (t == L - 1) ? next_value : values[t + 1];
        double nd = 1.0 - (double)dones[t];
        double delta = rewards[t] + gamma * nv * nd - values[t];
        gae = delta + gamma * lam * nd * gae;
        adv[t] = gae;
    }
    for (int64_t t = 0; t < L; ++t) ret[t] = adv[t] + values[t];
}

/* exported for tests */
double eo_det_log(double x) { return det_log(x); }
double eo_det_exp(double x) { return det_exp(x); }
double eo_det_sin2pi(double x) { return det_sin2pi(x); }
void eo_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed, uint32_t *out4)
{
    blk_t b = philox(c0, c1, c2, c3, seed);
    memcpy(out4, b.x, sizeof b.x);
}
