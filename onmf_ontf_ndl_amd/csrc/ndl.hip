// Motif-sampling Markov chains on a graph and their k x k adjacency patches on gfx950: the patch
// source of network dictionary learning (network_reconstruction_nx.py:108-134 tree_sample, :136-173
// glauber_gen_update, :175-199 RW_update, :265-278 Pivot_update, :280-313 chd_gen_mx with
// iterations=1, :315-329 get_patches_glauber).
//
// Integer / pointer-chasing work, latency-bound: ONE CHAIN PER WAVEFRONT.  Lane q holds emb[q] (the
// graph node that motif node q sits on, k <= 64); the sequential choices of a step are wave-uniform,
// the parallel parts (intersecting neighbour lists, the k*k edge tests of the patch) run across the
// lanes.  The graph is CSR with sorted rows (has_edge = binary search).  Throughput comes from many
// independent chains; a single chain reproduces the reference's one sequential chain.
//
// Randomness: Philox4x32-10, counter = (step, chain, slot / 4), key = seed - every draw is a pure
// function of (seed, chain, step, slot), so results do not depend on scheduling and a CPU
// restatement fed the same draws reproduces every embedding and patch bit for bit.  A uniform
// choice among n items is (draw * n) >> 32; the Metropolis test  rand() > deg_x / deg_y  of
// RW_update is the integer comparison  draw * deg_y > deg_x * 2^32.
// Slots of a step: 0 = motif node to update (Glauber) / proposed neighbour (random walk),
// 1 = choice among the common neighbours / Metropolis coin, 2 = uniform fallback node,
// 4 + i = neighbour choice for motif node i in tree_sample.
#include "common.h"

namespace {


__device__ __forceinline__ uint32_t chain_draw(uint64_t seed, uint64_t step, uint32_t chain, uint32_t slot) {
    uint32_t c[4] = {(uint32_t)step, (uint32_t)(step >> 32), chain, slot >> 2};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int rd = 0; rd < 10; ++rd) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const uint32_t e = slot & 3u;
    return e == 0 ? c[0] : e == 1 ? c[1] : e == 2 ? c[2] : c[3];
}

__device__ __forceinline__ int scaled(uint32_t x, int n) { return (int)(((uint64_t)x * (uint64_t)(uint32_t)n) >> 32); }

__device__ __forceinline__ bool has_edge(const int* __restrict__ row_ptr, const int* __restrict__ nbr, int a, int b) {
    int lo = row_ptr[a], hi = row_ptr[a + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int v = nbr[mid];
        if (v == b) return true;
        if (v < b) lo = mid + 1; else hi = mid;
    }
    return false;
}

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}

// RW_update (network_reconstruction_nx.py:175-199): uniform neighbour proposal, accepted with
// probability min(1, deg_x / deg_y); from an isolated node jump to a uniform node.  Wave-uniform.
__device__ __forceinline__ int rw_update(const int* __restrict__ row_ptr, const int* __restrict__ nbr, int n, int x,
                                         uint32_t d0, uint32_t d1) {
    const int sx = row_ptr[x], deg_x = row_ptr[x + 1] - sx;
    if (deg_x == 0) return scaled(d0, n);
    int y = nbr[sx + scaled(d0, deg_x)];
    const int deg_y = row_ptr[y + 1] - row_ptr[y];
    if ((uint64_t)d1 * (uint64_t)(uint32_t)deg_y > ((uint64_t)(uint32_t)deg_x << 32)) y = x;
    return y;
}

// mode 0: Glauber chain, 1: pivot chain, 2: initial embedding (tree_sample from a pivot, no patches)
__global__ __launch_bounds__(64) void ndl_chain_kernel(int n, const int* __restrict__ row_ptr,
                                                       const int* __restrict__ nbr, int k,
                                                       const int* __restrict__ motif, int mode, int steps,
                                                       int* __restrict__ emb, const int* __restrict__ pivots,
                                                       uint64_t seed, uint64_t step0, float* __restrict__ X, int ldx,
                                                       int* __restrict__ emb_trace) {
    const int lane = threadIdx.x;
    const uint32_t chain = blockIdx.x;
    // motif rows / columns of this lane's node as bit masks
    uint64_t in_mask = 0, out_mask = 0;
    if (lane < k) {
        for (int q = 0; q < k; ++q) {
            if (motif[q * k + lane] == 1) in_mask |= 1ull << q;
            if (motif[lane * k + q] == 1) out_mask |= 1ull << q;
        }
    }
    const bool motif_empty = __ballot(out_mask != 0) == 0;
    // find_parent (:100-106): lowest motif node with an edge into this one
    const int parent = in_mask ? (int)__builtin_ctzll(in_mask) : 0;
    int my = lane < k ? emb[(size_t)chain * k + lane] : 0;

    // tree_sample (:108-134) from pivot x: lane i ends up holding the node of motif node i
    auto tree_sample = [&](int x, uint64_t step) {
        const uint32_t mine = chain_draw(seed, step, chain, 4u + (uint32_t)lane);
        if (lane == 0) my = x;
        if (motif_empty) {
            if (lane >= 1 && lane < k) my = scaled(mine, n);
            return;
        }
        for (int i = 1; i < k; ++i) {
            const int p = __shfl(parent, i, 64);
            const int a = __shfl(my, p, 64);
            const uint32_t d = (uint32_t)__shfl((int)mine, i, 64);
            const int sa = row_ptr[a], deg = row_ptr[a + 1] - sa;
            const int y = deg > 0 ? nbr[sa + scaled(d, deg)] : a;
            if (lane == i) my = y;
        }
    };

    if (mode == 2) {
        const uint64_t step = ~0ull;
        const int x = pivots ? pivots[chain] : scaled(chain_draw(seed, step, chain, 0), n);
        tree_sample(x, step);
        if (lane < k) emb[(size_t)chain * k + lane] = my;
        return;
    }

    for (int s = 0; s < steps; ++s) {
        const uint64_t step = step0 + (uint64_t)s;
        const uint32_t d0 = chain_draw(seed, step, chain, 0);
        const uint32_t d1 = chain_draw(seed, step, chain, 1);
        const uint32_t d2 = chain_draw(seed, step, chain, 2);
        if (mode == 1) {
            // Pivot_update (:265-278): move the pivot by the random walk, re-embed the rest from it
            const int x0 = rw_update(row_ptr, nbr, n, __shfl(my, 0, 64), d0, d1);
            tree_sample(x0, step);
        } else if (k == 1) {
            my = rw_update(row_ptr, nbr, n, __shfl(my, 0, 64), d0, d1);      // (:144-152)
        } else {
            // glauber_gen_update (:155-172): resample one motif node among the common neighbours of
            // the nodes its motif neighbours sit on
            const int j = scaled(d0, k);
            const uint64_t nbh = shfl64(in_mask | out_mask, j);
            int y;
            if (nbh == 0) {
                y = scaled(d1, n);              // no constraint: uniform over all nodes
            } else {
                const int r0 = (int)__builtin_ctzll(nbh);
                const uint64_t others = nbh & ~(1ull << r0);
                const int a0 = __shfl(my, r0, 64);
                const int start = row_ptr[a0], deg = row_ptr[a0 + 1] - start;
                int total = 0;
                for (int base = 0; base < deg; base += 64) {
                    const int i = base + lane;
                    bool ok = i < deg;
                    const int c = ok ? nbr[start + i] : 0;
                    for (uint64_t m = others; m != 0; m &= m - 1) {
                        const int er = __shfl(my, (int)__builtin_ctzll(m), 64);
                        if (ok) ok = has_edge(row_ptr, nbr, er, c);
                    }
                    total += __popcll(__ballot(ok));
                }
                if (total == 0) {
                    y = scaled(d2, n);          // "Glauber move rejected": uniform node
                } else {
                    const int idx = scaled(d1, total);
                    int seen = 0;
                    y = 0;
                    for (int base = 0; base < deg; base += 64) {
                        const int i = base + lane;
                        bool ok = i < deg;
                        const int c = ok ? nbr[start + i] : 0;
                        for (uint64_t m = others; m != 0; m &= m - 1) {
                            const int er = __shfl(my, (int)__builtin_ctzll(m), 64);
                            if (ok) ok = has_edge(row_ptr, nbr, er, c);
                        }
                        const uint64_t bal = __ballot(ok);
                        const int cnt = __popcll(bal);
                        if (seen + cnt > idx) {
                            // the (idx - seen)-th set bit of bal
                            uint64_t mbits = bal;
                            for (int t = idx - seen; t > 0; --t) mbits &= mbits - 1;
                            y = __shfl(c, (int)__builtin_ctzll(mbits), 64);
                            break;
                        }
                        seen += cnt;
                    }
                }
            }
            if (lane == j) my = y;
        }
        // the patch: a2[q][r] = has_edge(emb[q], emb[r]) (:302-305), column chain*steps + s
        const size_t col = (size_t)chain * steps + s;
        if (emb_trace != nullptr && lane < k) emb_trace[col * k + lane] = my;
        for (int base = 0; base < k * k; base += 64) {      // (uniform trip count: every lane takes part in the shuffles)
            const int e = base + lane;
            const bool valid = e < k * k;
            const int q = valid ? e / k : 0, r = valid ? e - q * k : 0;
            const int a = __shfl(my, q, 64), b = __shfl(my, r, 64);
            if (valid) X[(size_t)e * ldx + col] = has_edge(row_ptr, nbr, a, b) ? 1.0f : 0.0f;
        }
    }
    if (lane < k) emb[(size_t)chain * k + lane] = my;
}

// Edge-wise accumulation of reconstructed patches (reconstruct_network, network_reconstruction_nx.py:
// 477-491): entry (q, r) of patch t belongs to the ordered node pair (emb_t[q], emb_t[r]); its values
// are averaged per pair.  Dense n x n sums and counts (the running mean of the reference is their
// quotient), float atomics: one thread per (entry, patch), patches fastest (coalesced reads of R).
__global__ __launch_bounds__(256) void ndl_edge_accumulate_kernel(int n, int k, int T, const int* __restrict__ emb_trace,
                                                                  const float* __restrict__ R, int ldr,
                                                                  float* __restrict__ sum, float* __restrict__ count) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int e = blockIdx.y;
    if (t >= T) return;
    const int q = e / k, r = e - q * k;
    const int a = emb_trace[(size_t)t * k + q], b = emb_trace[(size_t)t * k + r];
    const size_t at = (size_t)a * n + b;
    atomicAdd(sum + at, R[(size_t)e * ldr + t]);
    atomicAdd(count + at, 1.0f);
}

}  // namespace

int launch_ndl_edge_accumulate(int n, int k, int T, const int* emb_trace, const float* R, int ldr, float* sum,
                               float* count, hipStream_t s) {
    hipLaunchKernelGGL(ndl_edge_accumulate_kernel, dim3((T + 255) / 256, k * k), dim3(256), 0, s, n, k, T, emb_trace, R,
                       ldr, sum, count);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}

int launch_ndl_chains(int n, const int* row_ptr, const int* nbr, int k, const int* motif, int mode, int n_chains,
                      int steps, int* emb, const int* pivots, uint64_t seed, uint64_t step0, float* X, int ldx,
                      int* emb_trace, hipStream_t s) {
    hipLaunchKernelGGL(ndl_chain_kernel, dim3(n_chains), dim3(64), 0, s, n, row_ptr, nbr, k, motif, mode, steps, emb,
                       pivots, seed, step0, X, ldx, emb_trace);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}
