// hs_ring.hpp -- host side of ConsistentHash (components/load_balancer/strategies.py:336-433): md5 (RFC 1321), the ring of
// virtual nodes and the lookup.  Shared by the load-balancer pipeline (hs_lb.hip) and the general-graph engine (hs_graph.hip).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace hs {
namespace ring {

// host: md5 (RFC 1321) -- ConsistentHash._hash = int(hashlib.md5(key.encode()).hexdigest(), 16)
struct Md5 {
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    static uint32_t rotl(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t *p) {
        static const uint32_t T[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
            0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
            0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
            0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
            0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
            0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
            0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
            0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int Sh[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
        uint32_t w[16];
        for (int i = 0; i < 16; ++i) memcpy(&w[i], p + 4 * i, 4);        // little-endian host (x86-64)
        uint32_t A = a, Bv = b, Cv = c, Dv = d;
        for (int i = 0; i < 64; ++i) {
            const int rnd = i >> 4;
            uint32_t f; int g;
            switch (rnd) {
                case 0: f = Dv ^ (Bv & (Cv ^ Dv)); g = i; break;
                case 1: f = Cv ^ (Dv & (Bv ^ Cv)); g = (5 * i + 1) & 15; break;
                case 2: f = Bv ^ Cv ^ Dv; g = (3 * i + 5) & 15; break;
                default: f = Cv ^ (Bv | ~Dv); g = (7 * i) & 15; break;
            }
            const uint32_t tmp = Dv;
            Dv = Cv; Cv = Bv;
            Bv = Bv + rotl(A + f + T[i] + w[g], Sh[rnd][i & 3]);
            A = tmp;
        }
        a += A; b += Bv; c += Cv; d += Dv;
    }
    void digest(const char *msg, size_t len, uint8_t out[16]) {
        size_t off = 0;
        for (; off + 64 <= len; off += 64) block((const uint8_t *)msg + off);
        uint8_t tail[128] = {0};
        const size_t rem = len - off;
        memcpy(tail, msg + off, rem);
        tail[rem] = 0x80;
        const size_t tl = rem + 9 <= 64 ? 64 : 128;
        const uint64_t bits = (uint64_t)len * 8;
        memcpy(tail + tl - 8, &bits, 8);
        block(tail);
        if (tl == 128) block(tail + 64);
        memcpy(out, &a, 4); memcpy(out + 4, &b, 4); memcpy(out + 8, &c, 4); memcpy(out + 12, &d, 4);
    }
};

struct RingPoint { uint64_t hi, lo; int32_t backend, seq; };

inline void md5_u128(const char *msg, size_t len, uint64_t &hi, uint64_t &lo) {
    uint8_t dg[16];
    Md5 m;
    m.digest(msg, len, dg);
    hi = lo = 0;
    for (int i = 0; i < 8; ++i) { hi = (hi << 8) | dg[i]; lo = (lo << 8) | dg[8 + i]; }   // hexdigest read as one big integer
}

// first point with hash >= md5(key), else the first point: the reference's linear scan (strategies.py:423-433)
inline int32_t ring_select(const std::vector<RingPoint> &ring, const char *key, size_t len) {
    uint64_t hi, lo;
    md5_u128(key, len, hi, lo);
    size_t a = 0, b = ring.size();
    while (a < b) {
        const size_t m = (a + b) >> 1;
        const bool ge = ring[m].hi > hi || (ring[m].hi == hi && ring[m].lo >= lo);
        if (ge) b = m; else a = m + 1;
    }
    if (a == ring.size()) a = 0;                      // wrap around to the first node
    return ring[a].backend;
}

// ConsistentHash.add_backend for every backend in order (strategies.py:381-391): `vnodes` points md5(f"{name}:{i}") per backend,
// sorted by hash (list.sort is stable: insertion order on equal hashes).  names / name_off: the backends' names, concatenated.
inline std::vector<RingPoint> build_ring(const char *names, const int32_t *name_off, int n_backends, int vnodes) {
    std::vector<RingPoint> ring((size_t)n_backends * (size_t)(vnodes > 0 ? vnodes : 0));
    char key[256];
    size_t k = 0;
    for (int j = 0; j < n_backends && vnodes > 0; ++j) {
        const int nl = name_off[j + 1] - name_off[j];
        memcpy(key, names + name_off[j], (size_t)nl);
        for (int i = 0; i < vnodes; ++i, ++k) {
            const int m = snprintf(key + nl, sizeof(key) - (size_t)nl, ":%d", i);              // f"{backend.name}:{i}"
            md5_u128(key, (size_t)(nl + m), ring[k].hi, ring[k].lo);
            ring[k].backend = j; ring[k].seq = (int32_t)k;
        }
    }
    std::sort(ring.begin(), ring.end(), [](const RingPoint &x, const RingPoint &y) {
        if (x.hi != y.hi) return x.hi < y.hi;
        if (x.lo != y.lo) return x.lo < y.lo;
        return x.seq < y.seq;
    });
    return ring;
}

}  // namespace ring
}  // namespace hs
