/* c_abi_demo.c -- the engine through its C ABI only (no Python, no torch): plain C + the HIP runtime for buffers.
 *
 *   gcc -std=c99 -O2 examples/c_abi_demo.c -Iinclude -I/opt/rocm/include -Lbabyai_amd -lbbai_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/babyai_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_demo
 *   /tmp/c_abi_demo            # 4096 GoToLocal envs, 200 random steps + 100 expert-chosen steps, prints a state digest
 *
 * GoToLocal = Level_GoToLocal(room_size=8, num_dists=8) of babyai/levels/iclr19_levels.py:105-124.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include "bbai.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, bbai_last_error()); return 1; } } while (0)

int main(void) {
    const int64_t n = 4096;
    bbai_level_cfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.kind = 0;                      /* GoTo family */
    cfg.room_size = 8; cfg.num_rows = 1; cfg.num_cols = 1; cfg.num_dists = 8;
    cfg.check_reach = 1; cfg.instr = 1 /* go to */; cfg.target = 1 /* a random distractor */;
    CHECK(bbai_fill_layout(&cfg));

    bbai_env* env = NULL;
    CHECK(bbai_create(&cfg, n, 0, &env));
    uint64_t* seeds = (uint64_t*)malloc(n * sizeof(uint64_t));
    for (int64_t i = 0; i < n; ++i) seeds[i] = 1000 + (uint64_t)i;
    CHECK(bbai_seed(env, seeds, n));

    uint8_t *image, *dir, *done, *actions;
    float* reward;
    if (hipMalloc((void**)&image, n * BBAI_OBS_BYTES) || hipMalloc((void**)&dir, n) || hipMalloc((void**)&done, n) ||
        hipMalloc((void**)&actions, n) || hipMalloc((void**)&reward, n * sizeof(float))) return 2;
    CHECK(bbai_reset(env, image, dir, NULL));

    uint8_t* host_actions = (uint8_t*)malloc(n);
    float* host_reward = (float*)malloc(n * sizeof(float));
    uint8_t* host_done = (uint8_t*)malloc(n);
    uint32_t lcg = 12345;
    double reward_sum = 0.0;
    long episodes = 0;
    for (int t = 0; t < 200; ++t) {
        for (int64_t i = 0; i < n; ++i) { lcg = lcg * 1664525u + 1013904223u; host_actions[i] = (uint8_t)((lcg >> 24) % 7); }
        if (hipMemcpy(actions, host_actions, n, hipMemcpyHostToDevice)) return 2;
        CHECK(bbai_step(env, actions, image, dir, reward, NULL /* no f64 rewards */, done, 1 /* auto-reset */, NULL));
        if (hipMemcpy(host_reward, reward, n * sizeof(float), hipMemcpyDeviceToHost)) return 2;
        if (hipMemcpy(host_done, done, n, hipMemcpyDeviceToHost)) return 2;
        for (int64_t i = 0; i < n; ++i) { reward_sum += host_reward[i]; episodes += host_done[i]; }
    }
    /* performance knobs never change results (include/bbai.h bbai_set_option): consume finished envs inside the step kernel from here on */
    int64_t period = 0;
    CHECK(bbai_get_option(env, "lookahead_period", &period));
    CHECK(bbai_set_option(env, "consume_fused", 1));
    printf("look-ahead: one generator launch per %lld steps, ring of %lld levels per env\n", (long long)period, (long long)(2 * period));
    /* second phase: the reference's expert (babyai/bot.py) chooses the actions, on the device */
    long expert_episodes = 0, expert_solved = 0;
    for (int t = 0; t < 100; ++t) {
        CHECK(bbai_bot_act(env, NULL /* replan(None): the suggestion is what we step with */, actions, NULL));
        CHECK(bbai_step(env, actions, image, dir, reward, NULL, done, 1, NULL));
        if (hipMemcpy(host_reward, reward, n * sizeof(float), hipMemcpyDeviceToHost)) return 2;
        if (hipMemcpy(host_done, done, n, hipMemcpyDeviceToHost)) return 2;
        for (int64_t i = 0; i < n; ++i) { expert_episodes += host_done[i]; expert_solved += host_reward[i] > 0.0f; }
    }
    episodes += expert_episodes;
    uint64_t gave_up = 0, capacity = 0;
    CHECK(bbai_bot_stats(env, &gave_up, &capacity));
    printf("expert: episodes=%ld solved=%ld gave_up=%llu capacity=%llu\n", expert_episodes, expert_solved,
           (unsigned long long)gave_up, (unsigned long long)capacity);
    uint64_t resets = 0, failures = 0;
    CHECK(bbai_reset_count(env, &resets));
    CHECK(bbai_generator_failures(env, &failures));
    uint8_t* host_image = (uint8_t*)malloc(n * BBAI_OBS_BYTES);
    if (hipMemcpy(host_image, image, n * BBAI_OBS_BYTES, hipMemcpyDeviceToHost)) return 2;
    uint64_t digest = 1469598103934665603ull;
    for (int64_t i = 0; i < n * BBAI_OBS_BYTES; ++i) digest = (digest ^ host_image[i]) * 1099511628211ull;
    printf("envs=%lld steps=300 episodes_finished=%ld resets=%llu generator_failures=%llu reward_sum=%.6f obs_digest=%016llx\n",
           (long long)n, episodes, (unsigned long long)resets, (unsigned long long)failures, reward_sum, (unsigned long long)digest);
    bbai_destroy(env);
    return (resets == (uint64_t)n + (uint64_t)episodes && failures == 0) ? 0 : 3;
}
