// Cycle stamps of one wavefront of the env step kernel (phase by phase), for optimisation work.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DEM_STAMP -Iinclude -Imarl_amd/csrc \
//             tools/stamp_env.hip marl_amd/csrc/em_common.cpp -o tools/_stamp_env
// Run:    tools/_stamp_env [n_envs] [explore]
#include "../marl_amd/csrc/edge_env.hip"

#include <cstdlib>

int main(int argc, char** argv)
{
    const int64_t B = argc > 1 ? atoll(argv[1]) : 4096;
    const int explore = argc > 2 ? atoi(argv[2]) : 0;
    em_config c;
    std::memset(&c, 0, sizeof c);
    c.n_nodes = 10; c.episode_length = 1000; c.arrival_mode = 1; c.disable_env_exploration = !explore;
    const double caps[10] = {1.0e9, 1.2e9, 1.4e9, 1.6e9, 1.8e9, 2.0e9, 2.2e9, 2.4e9, 2.6e9, 3.0e9};
    for (int i = 0; i < 10; ++i) c.node_cpu_capacity[i] = caps[i];
    c.arrival_rate_lo = 5; c.arrival_rate_hi = 15; c.arrival_cycle_seconds = 20;
    c.workload_lo = 1; c.workload_hi = 10; c.data_lo = 0.5; c.data_hi = 5; c.deadline_lo = 0.05; c.deadline_hi = 0.5;
    c.bw_lo = 10; c.bw_hi = 100; c.net_latency_lo = 0.002; c.net_latency_hi = 0.01;
    c.w_latency = 0.5; c.w_energy = 0.3; c.w_violation = 0.2;
    c.min_time_step = 0.02; c.max_time_step = 0.2; c.initial_observation_window = 0.1;
    c.initial_epsilon = 1.0; c.final_epsilon = 0.05; c.decay_steps = 1000000;
    em_handle* h = nullptr;
    if (em_create(&c, B, 0, 7, 0, 0, &h)) { printf("create: %s\n", em_last_error()); return 1; }
    if (argc > 4 && em_set_lanes_per_env(h, atoi(argv[4]))) { printf("lanes: %s\n", em_last_error()); return 1; }
    if (argc > 3 && em_set_workgroup_waves(h, atoi(argv[3]))) { printf("waves: %s\n", em_last_error()); return 1; }
    const int n = 10;
    std::vector<int32_t> pl(B * n);
    std::vector<float> rs(B * n);
    for (int64_t i = 0; i < B * n; ++i) { pl[i] = (int)((i * 7 + i / 10) % n); rs[i] = 0.1f + 0.9f * (float)((i * 37) % 100) / 100.f; }
    int32_t* d_pl; float* d_rs; float* d_obs; uint64_t* d_act; double* d_rew; uint8_t* d_done;
    hipMalloc(&d_pl, B * n * 4); hipMalloc(&d_rs, B * n * 4); hipMalloc(&d_obs, B * n * (n + 4) * 4);
    hipMalloc(&d_act, B * 8); hipMalloc(&d_rew, B * 8); hipMalloc(&d_done, B);
    hipMemcpy(d_pl, pl.data(), B * n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_rs, rs.data(), B * n * 4, hipMemcpyHostToDevice);
    em_step_out out;
    std::memset(&out, 0, sizeof out);
    out.obs = d_obs; out.active = d_act; out.reward_seen = d_rew; out.done = d_done;
    if (em_reset(h, nullptr, 0, 0, &out, nullptr)) { printf("reset: %s\n", em_last_error()); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) em_step(h, d_pl, d_rs, nullptr, &out, nullptr);
    hipDeviceSynchronize();
    if (const char* cut = getenv("EM_CUT")) {
        // counter runs: the warm-up above brought the environments to a typical state; now every worker ends at stamp `cut`
        em_stamp_cut = atoi(cut);
        for (int i = 0; i < 1000; ++i) em_step(h, d_pl, d_rs, nullptr, &out, nullptr);
        hipDeviceSynchronize();
        printf("cut at stamp %d: 1000 launches\n", em_stamp_cut);
        return 0;
    }
    const int K = 500;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < K; ++i) em_step(h, d_pl, d_rs, nullptr, &out, nullptr);
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("B=%lld explore=%d: %.2f us/step\n", (long long)B, explore, ms * 1e3 / K);
    unsigned long long st[16];
    const int seq[10] = {0, 1, 2, 3, 4, 5, 6, 9, 10, 11};
    const char* names[10] = {"", "loads+philox+stage-in", "placement(par)", "chain", "reward", "(wait at barrier)",
                             "arrivals(par)", "arrival tail", "decay/outputs", "stage-out"};
    double acc[10] = {0}, sc[3] = {0};
    const int R = 20;
    for (int r = 0; r < R; ++r) {
        em_step(h, d_pl, d_rs, nullptr, &out, nullptr);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(st, HIP_SYMBOL(em_stamps), sizeof st);
        for (int i = 1; i < 10; ++i) acc[i] += (double)(st[seq[i]] - st[seq[i - 1]]);
        sc[0] += (double)((long long)(st[12] - st[0]));
        sc[1] += (double)(st[13] - st[12]);
        sc[2] += (double)(st[14] - st[13]);
    }
    if (argc > 5) {
        // fused mode: K steps per launch, stamps of the LAST step of the launch
        const int Kf = atoi(argv[5]);
        std::vector<int32_t> plk((size_t)Kf * B * n);
        std::vector<float> rsk((size_t)Kf * B * n);
        for (size_t i = 0; i < plk.size(); ++i) { plk[i] = pl[i % (B * n)]; rsk[i] = rs[i % (B * n)]; }
        int32_t* d_plk; float* d_rsk; double* d_rk; uint8_t* d_dk;
        hipMalloc(&d_plk, plk.size() * 4); hipMalloc(&d_rsk, rsk.size() * 4); hipMalloc(&d_rk, (size_t)Kf * B * 8); hipMalloc(&d_dk, (size_t)Kf * B);
        hipMemcpy(d_plk, plk.data(), plk.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_rsk, rsk.data(), rsk.size() * 4, hipMemcpyHostToDevice);
        em_step_out o2 = out;
        o2.reward_seen = d_rk; o2.done = d_dk;
        em_reset(h, nullptr, 0, 0, &out, nullptr);
        for (int i = 0; i < 10; ++i) acc[i] = 0;
        sc[0] = sc[1] = sc[2] = 0;
        hipEventRecord(e0, nullptr);
        const int reps = 3;
        for (int r = 0; r < reps; ++r) {
            if (em_step_many(h, Kf, EM_ACTIONS_DICT, d_plk, d_rsk, 0.5f, EM_HISTORY_REWARDS, &o2, nullptr)) { printf("many: %s\n", em_last_error()); return 1; }
        }
        hipEventRecord(e1, nullptr);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        printf("fused K=%d: %.2f us/step\n", Kf, ms * 1e3 / (reps * Kf));
        hipMemcpyFromSymbol(st, HIP_SYMBOL(em_stamps), sizeof st);
        for (int i = 1; i < 10; ++i) acc[i] = (double)(st[seq[i]] - st[seq[i - 1]]) * R;
        sc[1] = (double)(st[13] - st[12]) * R; sc[2] = (double)(st[14] - st[13]) * R; sc[0] = 0;
        printf("  next step's STAMP(0) - this STAMP(11) not shown; loop period is the sum below\n");
    }
    double tot = 0;
    for (int i = 1; i < 10; ++i) { printf("  %-22s %8.0f cycles\n", names[i], acc[i] / R); tot += acc[i] / R; }
    printf("  %-22s %8.0f cycles (s_memtime ticks)\n", "worker total", tot);
    printf("  scalar wavefront: starts %+.0f after worker 0, rate+window %.0f, poisson %.0f cycles\n", sc[0] / R, sc[1] / R, sc[2] / R);
    em_destroy(h);
    return 0;
}
