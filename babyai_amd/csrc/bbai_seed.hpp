// bbai_seed.hpp -- env seeding: seed int -> MT19937 state, bit-compatible with
// `MiniGridEnv.seed(s)` = gym.utils.seeding.np_random(s) (gym <= 0.21, the API generation the
// reference's call sites require: scripts/train_rl.py:59, babyai/evaluate.py:105-106):
//   sha512(str(seed)) -> first 8 bytes, little-endian -> uint32 words (low first, leading zero
//   words dropped) -> numpy RandomState.seed(words) = MT19937 init_by_array.
// Runs per lane in k_seed (8 bytes per env cross PCIe, not the 2.5 KB state) and on the host in the test harness.
#pragma once
#include <stdint.h>
#include <string.h>
#include "bbai_types.hpp"

namespace bbai {

namespace sha512_detail {
BB_HD uint64_t sha_k(int t) {
    constexpr uint64_t K[80] = {
    0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
    0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
    0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
    0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
    0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
    0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
    0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
    0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
    0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
    0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
    0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
    0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
    0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
    0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
    0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};
    return K[t];
}
BB_HD uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
}  // namespace sha512_detail

// First 8 bytes of sha512(msg) for a short message (< 112 bytes => one block).
BB_HD void sha512_first8(const char* msg, int len, uint8_t out8[8]) {
    using namespace sha512_detail;
    // message schedule as a 16-word ring (the message is < 112 bytes: one block, big-endian words)
    uint64_t w[16];
    for (int t = 0; t < 16; ++t) w[t] = 0;
    for (int i = 0; i < len; ++i) w[i >> 3] |= (uint64_t)(uint8_t)msg[i] << (56 - 8 * (i & 7));
    w[len >> 3] |= (uint64_t)0x80 << (56 - 8 * (len & 7));
    w[15] = (uint64_t)len * 8;
    uint64_t a = 0x6a09e667f3bcc908ULL, b = 0xbb67ae8584caa73bULL, c = 0x3c6ef372fe94f82bULL, d = 0xa54ff53a5f1d36f1ULL;
    uint64_t e = 0x510e527fade682d1ULL, f = 0x9b05688c2b3e6c1fULL, g = 0x1f83d9abfb41bd6bULL, h = 0x5be0cd19137e2179ULL;
    const uint64_t a0 = a;
#pragma unroll
    for (int t = 0; t < 80; ++t) {
        if (t >= 16) {
            const uint64_t w15 = w[(t - 15) & 15], w2 = w[(t - 2) & 15];
            const uint64_t s0 = rotr(w15, 1) ^ rotr(w15, 8) ^ (w15 >> 7);
            const uint64_t s1 = rotr(w2, 19) ^ rotr(w2, 61) ^ (w2 >> 6);
            w[t & 15] = w[t & 15] + s0 + w[(t - 7) & 15] + s1;
        }
        uint64_t S1 = rotr(e, 14) ^ rotr(e, 18) ^ rotr(e, 41);
        uint64_t ch = (e & f) ^ (~e & g);
        uint64_t t1 = h + S1 + ch + sha_k(t) + w[t & 15];
        uint64_t S0 = rotr(a, 28) ^ rotr(a, 34) ^ rotr(a, 39);
        uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint64_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    uint64_t h0 = a0 + a;   // only the first state word is needed
    for (int i = 0; i < 8; ++i) out8[i] = (uint8_t)(h0 >> (8 * (7 - i)));
}

// numpy RandomState.seed(key) = MT19937 init_by_array(key), key_len 1 or 2.  `mt` may be global memory written by one
// lane: the chain value is carried in a register, and the seed-independent init_genrand(19650218) pass is folded into
// the first key pass (each word is written once there and read + written once by the second pass).
BB_HD void mt_init_by_array(uint32_t* mt, const uint32_t* key, int key_len) {
    uint32_t init = 19650218u;                   // init_genrand word i, generated on the fly
    uint32_t prev = init;                        // mt[i - 1] of the running recurrence (mt[0] = init before any wrap)
    int j = 0;
    for (int i = 1; i < 624; ++i) {              // first 623 of the max(624, key_len) = 624 key steps
        init = 1812433253u * (init ^ (init >> 30)) + (uint32_t)i;
        prev = (init ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        mt[i] = prev;
        if (++j >= key_len) j = 0;
    }
    // i wrapped: mt[0] = mt[623]; the 624th key step rewrites mt[1]
    prev = (mt[1] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    mt[1] = prev;
    int i = 2;
    for (int k = 623; k; --k) {
        prev = (mt[i] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
        mt[i] = prev;
        if (++i >= 624) i = 1;                   // (mt[0] = mt[623] is overwritten below; prev already carries it)
    }
    mt[0] = 0x80000000u;
}

// env.seed(seed): fills the 624-word state; the output index starts at 624 (twist first).
BB_HD void seed_env(uint64_t seed, uint32_t* mt) {
    char buf[24], rev[24];
    int len = 0;                                 // str(seed): decimal digits, most significant first
    do { rev[len++] = (char)('0' + (int)(seed % 10)); seed /= 10; } while (seed);
    for (int i = 0; i < len; ++i) buf[i] = rev[len - 1 - i];
    uint8_t d[8];
    sha512_first8(buf, len, d);
    uint32_t lo = (uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[2] << 16) | ((uint32_t)d[3] << 24);
    uint32_t hi = (uint32_t)d[4] | ((uint32_t)d[5] << 8) | ((uint32_t)d[6] << 16) | ((uint32_t)d[7] << 24);
    uint32_t key[2] = {lo, hi};
    int n = hi != 0 ? 2 : 1;            // _int_list_from_bigint drops leading zero words ([0] for 0)
    mt_init_by_array(mt, key, n);
}

}  // namespace bbai
