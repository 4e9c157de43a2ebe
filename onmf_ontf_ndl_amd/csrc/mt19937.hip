// NumPy's legacy global generator (MT19937) continued on the GPU: np.random.rand(n) as the reference draws
// its start code H0 = np.random.rand(r, b) in every solve (src/onmf.py:245-246), bit for bit, written
// straight into device memory as float32 (the value the host path gets by casting the float64 draw).
//
// The recurrence on the raw words  y[n+624] = y[n+397] ^ mix(y[n], y[n+1])  lets at most 227 words be
// computed at a time, so one stream cannot be produced by more than one workgroup - unless the workgroups
// start from windows far apart.  MT19937 is linear over GF(2): with g_J = x^J mod phi (phi: the minimal
// polynomial of the one-word shift; the host computes the table once, onmf_ontf_ndl_amd/mt_jump.py),
//     y[J + k] = XOR over the set bits i of g_J of y[i + k]            (k >= 1; k = 0: top bit only)
// i.e. a window J words ahead is a correlation of 19937 bits against the first 20 560 words of the stream.
//   mt_head_kernel   one workgroup: those first words, from the generator's current block
//   mt_jump_kernel   one workgroup per chunk: its window from the table row and the head words (all in LDS)
//   mt_fill_kernel   one workgroup per chunk of 420 blocks: twist (three phases of <= 227 independent words),
//                    tempering, pairs of words -> 53-bit doubles -> float32, and the block the generator
//                    stops in (NumPy gets it back as its new state)
#include "common.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace {

constexpr int MTN = 624, MTM = 397, MT_DEGREE = 19937;
constexpr int MT_CHUNK_BLOCKS = 210;
constexpr long long MT_CHUNK_WORDS = (long long)MTN * MT_CHUNK_BLOCKS;
constexpr int MT_HEAD_BLOCKS = 33;                    // 33 * 624 = 20 592 >= 19 937 + 623
constexpr int MT_HEAD_WORDS = MT_HEAD_BLOCKS * MTN;
constexpr int MT_THREADS = 256;

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((b & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// nxt = the 624 words after cur (both in LDS); every thread of the workgroup calls it
__device__ __forceinline__ void mt_twist(const uint32_t* cur, uint32_t* nxt) {
    const int tid = threadIdx.x;
    if (tid < 227) nxt[tid] = mt_mix(cur[tid], cur[tid + 1], cur[tid + MTM]);
    __syncthreads();
    if (tid < 227) nxt[227 + tid] = mt_mix(cur[227 + tid], cur[228 + tid], nxt[tid]);
    __syncthreads();
    if (tid < 169) nxt[454 + tid] = mt_mix(cur[454 + tid], cur[455 + tid], nxt[227 + tid]);
    if (tid == 255) nxt[623] = mt_mix(cur[623], nxt[0], nxt[396]);
    __syncthreads();
}

__global__ __launch_bounds__(MT_THREADS) void mt_head_kernel(const uint32_t* __restrict__ block0,
                                                              uint32_t* __restrict__ Y) {
    __shared__ uint32_t buf[2][MTN];
    const int tid = threadIdx.x;
    for (int k = tid; k < MTN; k += MT_THREADS) {
        buf[0][k] = block0[k];
        Y[k] = block0[k];
    }
    __syncthreads();
    for (int q = 1; q < MT_HEAD_BLOCKS; ++q) {
        const uint32_t* cur = buf[(q - 1) & 1];
        uint32_t* nxt = buf[q & 1];
        mt_twist(cur, nxt);
        for (int k = tid; k < MTN; k += MT_THREADS) Y[q * MTN + k] = nxt[k];
    }
}

// blocks[c] (c = 1 + blockIdx.x) = y[c * chunk .. c * chunk + 623] from list c - 1: the positions i of the set
// bits of x^(c * chunk - 1) mod phi, sorted by i & 3 into four classes and stored as i >> 2.  Within a class every
// window starts at the same offset from a 16-byte boundary, so a thread that owns four consecutive words of
// the (class-shifted) sum reads whole aligned quads: acc_a[k'] ^= ys[4m + k'] for k' = 4t .. 4t + 3, and the
// window is win[k] = XOR_a acc_a[k + a].  Each class is padded to a multiple of 8 entries with MT_HEAD_WORDS / 4
// (a zeroed stretch behind the head words); eight quads are in flight at a time.  The list is split over
// MT_JUMP_GROUPS groups of three wavefronts (157 of 192 lanes own a quad).
constexpr int MT_LIST_STRIDE = 10240;                 // entries per list (19937 bits, half of them set, and room)
constexpr int MT_JUMP_GROUPS = 2;
constexpr int MT_JUMP_THREADS = 192 * MT_JUMP_GROUPS;
constexpr int MT_QUADS = 157;                         // ceil((624 + 3) / 4)
constexpr int MT_ZERO_WORDS = 1024;

__global__ __launch_bounds__(MT_JUMP_THREADS) void mt_jump_kernel(const uint32_t* __restrict__ Y,
                                                                   const uint16_t* __restrict__ lists,
                                                                   const int* __restrict__ counts,
                                                                   uint32_t* __restrict__ blocks) {
    extern __shared__ __align__(16) uint32_t lds[];
    uint32_t* ys = lds;                                   // MT_HEAD_WORDS + MT_ZERO_WORDS zeros
    uint32_t* win = lds + MT_HEAD_WORDS + MT_ZERO_WORDS;  // 640
    uint16_t* lst = reinterpret_cast<uint16_t*>(win + 640);           // the chunk's list, later the partial sums
    const int tid = threadIdx.x;
    const int grp = tid / 192, t = tid % 192;
    const int* cnt = counts + 4 * blockIdx.x;             // entries per class, multiples of 8
    const int total = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    const uint16_t* mine = lists + (size_t)blockIdx.x * MT_LIST_STRIDE;
    for (int k = tid; k < MT_HEAD_WORDS; k += MT_JUMP_THREADS) ys[k] = Y[k];
    for (int k = tid; k < MT_ZERO_WORDS; k += MT_JUMP_THREADS) ys[MT_HEAD_WORDS + k] = 0u;
    for (int k = tid; k < total / 2; k += MT_JUMP_THREADS)
        reinterpret_cast<uint32_t*>(lst)[k] = reinterpret_cast<const uint32_t*>(mine)[k];
    __syncthreads();
    uint4 acc[4];
    const bool owner = t < MT_QUADS;
    const uint4* yq = reinterpret_cast<const uint4*>(ys) + (owner ? t : 0);
    int seg = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        acc[a] = make_uint4(0u, 0u, 0u, 0u);
        const int octs = cnt[a] / 8;                      // this group's share, in eights
        const int lo = seg + 8 * (octs * grp / MT_JUMP_GROUPS), hi = seg + 8 * (octs * (grp + 1) / MT_JUMP_GROUPS);
        for (int e = lo; e < hi; e += 8) {
            const uint4 pk = *reinterpret_cast<const uint4*>(lst + e);  // same address in every lane of the group
            const uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w};
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = yq[(w[u >> 1] >> (16 * (u & 1))) & 0xffff];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[a].x ^= v[u].x;
                acc[a].y ^= v[u].y;
                acc[a].z ^= v[u].z;
                acc[a].w ^= v[u].w;
            }
        }
        seg += cnt[a];
    }
    __syncthreads();                                      // the list is read; its room takes the partial sums
    uint4* part = reinterpret_cast<uint4*>(lst);          // [group][class][MT_QUADS + 3 quads]
    constexpr int PQ = MT_QUADS + 3;
    if (owner) {
#pragma unroll
        for (int a = 0; a < 4; ++a) part[(grp * 4 + a) * PQ + t] = acc[a];
    }
    __syncthreads();
    const uint32_t* pw = reinterpret_cast<const uint32_t*>(part);
    for (int k = tid; k < MTN; k += MT_JUMP_THREADS) {
        uint32_t x = 0;
#pragma unroll
        for (int g = 0; g < MT_JUMP_GROUPS; ++g)
#pragma unroll
            for (int a = 0; a < 4; ++a) x ^= pw[(g * 4 + a) * PQ * 4 + k + a];
        win[k] = x;
    }
    __syncthreads();
    // the window is one word early: [y[J-1] (top bit right), y[J], ..., y[J+622]]; the block wanted is
    // y[J .. J+623], whose last word follows from the recurrence
    uint32_t* out = blocks + (size_t)(blockIdx.x + 1) * MTN;
    for (int k = tid; k < MTN - 1; k += MT_JUMP_THREADS) out[k] = win[k + 1];
    if (tid == 0) out[MTN - 1] = mt_mix(win[0], win[1], win[MTM]);
}

struct MtFillArgs {
    const uint32_t* blocks;      // n_chunks x 624: the raw block each chunk starts with
    long long base;              // index in the stream of raw words of chunk 0's first word (0 = first word of
                                 // the block NumPy's state held when the draw began)
    long long first;             // index of the first word the draw consumes (NumPy's `pos`)
    long long n;                 // doubles to draw
    float* out;                  // n floats
    uint32_t* final_block;       // 624 words + 1: the block holding the last consumed word, and pos within it
    uint32_t* carry;             // 624 words: the block after the last chunk's last (the next round starts there)
    int n_chunks;
    int need_carry;              // another round follows: the last chunk runs to its end and leaves `carry`
};

__global__ __launch_bounds__(MT_THREADS) void mt_fill_kernel(const MtFillArgs a) {
    __shared__ uint32_t buf[2][MTN];
    const int tid = threadIdx.x;
    const int c = blockIdx.x;
    for (int k = tid; k < MTN; k += MT_THREADS) buf[0][k] = a.blocks[(size_t)c * MTN + k];
    __syncthreads();
    const long long end = a.first + 2 * a.n;                   // one past the last word consumed
    const long long last_block_at = a.n > 0 ? (end - 1) / MTN * MTN : -1;
    long long base = a.base + (long long)c * MT_CHUNK_WORDS;
    for (int q = 0; q < MT_CHUNK_BLOCKS; ++q, base += MTN) {
        const uint32_t* cur = buf[q & 1];
        uint32_t* nxt = buf[(q + 1) & 1];
        const bool wanted = base < end && base + MTN > a.first;
        if (base >= end && !(a.need_carry && c == a.n_chunks - 1)) break;      // nothing left for this chunk (uniform)
        mt_twist(cur, nxt);                                     // (also gives the pair that straddles the blocks)
        if (wanted) {
            // first words of pairs sit at stream offsets of the same parity as a.first
            const int par = (int)((a.first - base) & 1);
            for (int i = tid; 2 * i + par < MTN; i += MT_THREADS) {
                const int k = 2 * i + par;
                const long long gi = base + k;
                if (gi < a.first) continue;
                const long long e = (gi - a.first) >> 1;
                if (e >= a.n) continue;
                const uint32_t w1 = mt_temper(cur[k]);
                const uint32_t w2 = mt_temper(k + 1 < MTN ? cur[k + 1] : nxt[0]);
                const double v = ((double)(w1 >> 5) * 67108864.0 + (double)(w2 >> 6)) * (1.0 / 9007199254740992.0);
                a.out[e] = (float)v;
            }
            if (base == last_block_at) {
                for (int k = tid; k < MTN; k += MT_THREADS) a.final_block[k] = cur[k];
                if (tid == 0) a.final_block[MTN] = (uint32_t)(end - 1 - base) + 1u;
            }
        }
        // nxt's readers (the straddling pair) are done before it becomes `cur` and the old `cur` is overwritten
        __syncthreads();
    }
    if (a.need_carry && c == a.n_chunks - 1) {
        const uint32_t* fin = buf[MT_CHUNK_BLOCKS & 1];
        for (int k = tid; k < MTN; k += MT_THREADS) a.carry[k] = fin[k];
    }
}

}  // namespace

// Device buffers of the generator (onmf_ctx::mt), in 32-bit words:
// [lists rows * MT_LIST_STRIDE / 2 | counts rows * 4 | Y head | blocks (rows+1)*624 | final 640 | carry 624]
static size_t mt_list_words(int rows) { return (size_t)rows * (MT_LIST_STRIDE / 2); }
static size_t mt_count_words(int rows) { return (size_t)rows * 4; }
int launch_numpy_rand(onmf_ctx* c, const uint32_t* key_host, int pos, size_t n, float* out, uint32_t* key_out_host,
                      int* pos_out, hipStream_t s) {
    const int rows = c->mt_rows;
    const int per_round = rows + 1;                         // chunks one round can start at once
    const size_t off_lists = 0;
    const size_t off_counts = off_lists + mt_list_words(rows);
    const size_t off_y = off_counts + mt_count_words(rows);
    const size_t off_blocks = off_y + MT_HEAD_WORDS;
    const size_t off_final = off_blocks + (size_t)per_round * MTN;
    const size_t off_carry = off_final + 640;
    const size_t total = off_carry + MTN;
    if (c->mt.bytes < total * sizeof(uint32_t))
        return onmf_fail(c, ONMF_ERR_ARG, "numpy_rand: the generator's buffers are not set up");
    uint32_t* base = (uint32_t*)c->mt.p;
    ONMF_TRY(onmf_reserve_pinned(c, 1024 * sizeof(uint32_t)));
    uint32_t* pin = (uint32_t*)c->pinned;
    memcpy(pin, key_host, MTN * sizeof(uint32_t));
    HIP_TRY(c, hipMemcpyAsync(base + off_blocks, pin, MTN * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    static LdsLimit lds_limit;
    const size_t jump_lds = (size_t)(MT_HEAD_WORDS + MT_ZERO_WORDS + 640) * sizeof(uint32_t) + MT_LIST_STRIDE * sizeof(uint16_t);
    const long long end = (long long)pos + 2 * (long long)n;
    const long long chunks_total = n > 0 ? (end + MT_CHUNK_WORDS - 1) / MT_CHUNK_WORDS : 0;
    long long done = 0;
    while (done < chunks_total) {
        const int nc = (int)std::min<long long>(per_round, chunks_total - done);
        if (done > 0)       // this round starts where the last one's carry left off
            HIP_TRY(c, hipMemcpyAsync(base + off_blocks, base + off_carry, MTN * sizeof(uint32_t),
                                      hipMemcpyDeviceToDevice, s));
        if (nc > 1) {
            hipLaunchKernelGGL(mt_head_kernel, dim3(1), dim3(MT_THREADS), 0, s, base + off_blocks, base + off_y);
            ONMF_TRY(onmf_raise_lds_limit(c, lds_limit, (const void*)mt_jump_kernel, jump_lds));
            hipLaunchKernelGGL(mt_jump_kernel, dim3(nc - 1), dim3(MT_JUMP_THREADS), jump_lds, s, base + off_y,
                               reinterpret_cast<const uint16_t*>(base + off_lists),
                               reinterpret_cast<const int*>(base + off_counts), base + off_blocks);
        }
        MtFillArgs a;
        a.blocks = base + off_blocks;
        a.base = done * MT_CHUNK_WORDS;
        a.first = pos;
        a.n = (long long)n;
        a.out = out;
        a.final_block = base + off_final;
        a.carry = base + off_carry;
        a.n_chunks = nc;
        a.need_carry = done + nc < chunks_total;
        hipLaunchKernelGGL(mt_fill_kernel, dim3(nc), dim3(MT_THREADS), 0, s, a);
        if (hipGetLastError() != hipSuccess) return onmf_fail(c, ONMF_ERR_HIP, "numpy_rand: launch failed");
        done += nc;
    }
    if (n == 0) {
        memcpy(key_out_host, key_host, MTN * sizeof(uint32_t));
        *pos_out = pos;
        return ONMF_OK;
    }
    HIP_TRY(c, hipMemcpyAsync(pin, base + off_final, (MTN + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    memcpy(key_out_host, pin, MTN * sizeof(uint32_t));
    *pos_out = (int)pin[MTN];
    return ONMF_OK;
}

int numpy_rand_setup(onmf_ctx* c, const uint32_t* table_host, int rows, hipStream_t s) {
    const size_t total = mt_list_words(rows) + mt_count_words(rows) + MT_HEAD_WORDS + (size_t)(rows + 1) * MTN + 640 + MTN;
    ONMF_TRY(onmf_reserve(c, c->mt, total * sizeof(uint32_t), "MT19937 buffers"));
    if (rows > 0) {
        // the rows' set bits as lists of positions (what the jump kernel walks): by position & 3, then
        // ascending, stored as position >> 2, each class padded to a multiple of 8
        std::vector<uint16_t> lists((size_t)rows * MT_LIST_STRIDE, (uint16_t)(MT_HEAD_WORDS / 4));
        std::vector<int> counts(mt_count_words(rows), 0);
        for (int j = 0; j < rows; ++j) {
            int n = 0;
            for (int a = 0; a < 4; ++a) {
                const int start = n;
                for (int i = a; i < MT_DEGREE; i += 4)
                    if ((table_host[(size_t)j * MTN + (i >> 5)] >> (i & 31)) & 1u) {
                        if (n >= MT_LIST_STRIDE - 32)
                            return onmf_fail(c, ONMF_ERR_ARG, "numpy_rand: table row %d has too many bits", j);
                        lists[(size_t)j * MT_LIST_STRIDE + n++] = (uint16_t)(i >> 2);
                    }
                n = start + ((n - start + 7) & ~7);
                counts[4 * j + a] = n - start;
            }
        }
        uint32_t* base = (uint32_t*)c->mt.p;
        HIP_TRY(c, hipMemcpyAsync(base, lists.data(), lists.size() * sizeof(uint16_t), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(base + mt_list_words(rows), counts.data(), counts.size() * sizeof(int),
                                  hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    c->mt_rows = rows;
    return ONMF_OK;
}

long long numpy_rand_chunk_words() { return MT_CHUNK_WORDS; }
