// bbai_kernels.hpp -- what the engine's translation units share on the device side: the look-ahead ring's addressing, the refill list's
// shape, the lane-group context of the generators.  (bbai_engine.hip: every kernel but one; bbai_genlane.hip: k_pregen_lane, compiled
// with its own flags -- see there.)
#pragma once
#include <hip/hip_runtime.h>
#include "bbai_types.hpp"

namespace bbai {

constexpr int META_U32 = 32;            // uint32 per window buffer's meta line: [0] = M when > 1 (atomicMax)
constexpr int SHARDS = 64;              // cache lines the reset total is spread over (k_step: shard = block & 63)
constexpr int SHARD_U64 = 16;           // uint64 per shard: one 128-byte line each
enum : int { FLOW_REFILLED = 0 /* windows whose refill has landed */, FLOW_GATE_TIMEOUTS = 1, FLOW_PROBE = 2 /* probe_stream's flag */, FLOW_GEN_FAILURES = 3 /* levels the generator gave up on */, FLOW_WORDS = 16 };
constexpr int GEN_COUNT_U32 = 32;       // uint32 per sub-list counter of the refill list: one 128-byte line each
__host__ __device__ __forceinline__ int64_t gen_sublist_cap(int64_t n) { return ((n + 63) / 64 + SHARDS - 1) / SHARDS * 64; }      // entries a sub-list can get: its waves x 64

// The look-ahead ring is ENV-MAJOR: entry (slot, env) of next_rec / next_hot / next_obs is number env * depth + slot -- an env's D levels
// lie together.  (Slot-major, rounds 1-4a, put the 64 envs of a stepping wave into up to 64 regions n * rec_bytes apart as soon as
// the live records are ring slots: profiles/r04/inplace_ring_depth_ab.jsonl, k_step 0.029 -> 0.035 ms from D = 5 to D = 65.)
__device__ __forceinline__ int64_t ring_at(int slot, int64_t env, int depth) { return env * depth + slot; }

// One env per group of G lanes, 64 / G envs per wavefront (bbai_gen.hpp "Execution model").  sync() orders the group's LDS
// accesses: it is reached under divergent control flow (the groups of a wave are in different places of the generator),
// so it is a wave-local fence, never a workgroup barrier -- the workgroup is one wave.
template <int G>
struct GroupCtx {
    static constexpr int kLanes = G;
    __device__ __forceinline__ int lane() const { return (int)threadIdx.x & (G - 1); }
    __device__ __forceinline__ int nlanes() const { return G; }
    __device__ __forceinline__ void sync() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    __device__ __forceinline__ uint32_t shfl(uint32_t v, int src) const { return __shfl(v, src, G); }
    __device__ __forceinline__ uint32_t shfl_up1(uint32_t v) const { uint32_t t = __shfl_up(v, 1, G); return lane() == 0 ? 0u : t; }
    __device__ __forceinline__ uint32_t shfl_down1(uint32_t v) const { uint32_t t = __shfl_down(v, 1, G); return lane() == G - 1 ? 0u : t; }
    __device__ __forceinline__ unsigned long long ballot(bool p) const {     // the group's share of the wave's ballot: bit k = lane k of the group
        const unsigned long long b = __ballot(p);
        constexpr unsigned long long m = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
        return (b >> ((int)threadIdx.x & ~(G - 1) & 63)) & m;
    }
    __device__ __forceinline__ bool any(bool p) const {
        const unsigned long long b = __ballot(p);
        constexpr unsigned long long m = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
        return ((b >> ((int)threadIdx.x & ~(G - 1) & 63)) & m) != 0ull;
    }
};

// k_pregen_lane's launch arguments: filled by bbai_engine.hip (launch_pregen_lane), launched by bbai_genlane.hip (bbai_lane_launch)
struct LaneLaunch {
    LevelCfg cfg; int64_t n; uint8_t* next_rec; Hot* next_hot; uint32_t* mt; uint32_t* mtt; uint8_t* mtpar; int32_t* mti;
    const int32_t* gen_list; const uint32_t* gen_count; int depth; uint8_t* pending; const uint8_t* first_slot;
    unsigned long long* fails; uint8_t* next_obs; const uint8_t* tmpl; int lane_words; unsigned blocks; hipStream_t stream;
};
void bbai_lane_launch(const LaneLaunch& a);

}  // namespace bbai
