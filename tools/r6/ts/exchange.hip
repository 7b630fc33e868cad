// exchange.hip — one exchange point of a column-sharded decode step as ONE small kernel, without a collective library (gfx950; peers
// reached through memory the caller mapped with hipIpcOpenMemHandle: stores travel over xGMI).
//
// What it replaces: `all_gather_into_tensor` of the shard outputs + the permuting copy that restores the reference's column order
// (hqq_amd/shard.py; SURVEY.md section 8e — the reference itself has no multi-GPU path for HQQLinear.forward, quantize.py:880-898).
// At one activation row an exchange moves N/P x 2 bytes per rank (1-7 KiB at the 70B shapes): a collective library's launch + protocol
// latency is several times the 4-8 us GEMV it follows.  Here every rank
//   1. stores its slice straight into every peer's FULL output row, at the columns the reference order gives it: rank r's packed-row
//      block yields, per slab s, the columns s N/per + [r n', (r + 1) n'), n' = N / (per P) — so nothing is permuted afterwards,
//   2. makes those stores visible at system scope and writes the exchange's GENERATION into its flag word of that peer's flag block (a
//      plain 4-byte store, the one thing that certainly works over xGMI — no remote atomics).  The generation is the number of times
//      this point has been used, counted on the device (the kernel's workgroups draw tickets from a local counter: ticket / world), so
//      it also counts under hipGraph replay; every rank issues the same sequence of points, so every rank counts the same,
//   3. waits, in the workgroup that served its own row, until all P flag words of its own block have reached the generation.  Flags are
//      never lowered: a flag that arrives late (after a wait gave up) is one generation behind the next use of the point and cannot
//      satisfy it (round 3 lowered 0 / 1 flags: a late flag stayed raised and let the next exchange through before its data — advisor).
// When the kernel has finished, this rank's full rows are complete: the next kernel in stream order may read them (kernel boundaries
// order memory at device scope, as for any stream-ordered pair of kernels).
//
// Re-use rule (the caller's): consecutive exchanges on a stream alternate between AT LEAST TWO points (flag block + rows).  Then a peer
// can raise its flag for the next use of a point only after this rank has raised a flag for a LATER point, i.e. after this kernel has
// finished here — flags are never raised onto a block that is still being waited on, and rows are never overwritten before the kernel
// that reads them has run.  A decoder block has four points (q|k|v, o, gate|up, down).
// A wait gives up after spin_limit polls: it writes 1 + the index of the missing rank to *status (sticky until the caller's collective
// reset) and returns (outputs of that exchange undefined, reported, never a hang).
// Flag block of a point (local memory of each rank): HQQ_EXCHANGE_MAX_RANKS flag words + the launch-ticket word.
#include "hqq_common.h"

namespace hqq {

constexpr int XG_MAXL = HQQ_GEMV_MAX_GROUP;
constexpr int XG_MAXP = HQQ_EXCHANGE_MAX_RANKS;

struct XgArgs {
  const uint16_t* y_loc[XG_MAXL];          // this rank's [M, n_loc[j]] outputs, local (slab-major) order inside a row
  uint16_t* full[XG_MAXP][XG_MAXL];        // rank p's [M, n_loc[j] * world] rows of layer j (p == rank: local memory)
  uint32_t* flags[XG_MAXP];                // rank p's flag block of this point: [world] words
  uint32_t* status;
  int n_loc[XG_MAXL];
  int n_layers, per, world, rank, M;     // M activation rows (1..HQQ_EXCHANGE_MAX_ROWS): y_loc[j] is [M, n_loc[j]], a full row set [M, n_loc[j] * world]
  uint32_t spin_limit;
};

__global__ __launch_bounds__(256) void exchange_kernel(const XgArgs a) {
  __shared__ uint32_t gen_s;
  const int tid = threadIdx.x;
  // this launch's generation: its `world` workgroups draw tickets n world .. n world + world - 1 from the point's local counter (order-free)
  if (tid == 0) gen_s = __hip_atomic_fetch_add(a.flags[a.rank] + XG_MAXP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / static_cast<uint32_t>(a.world) + 1u;
  __syncthreads();
  const uint32_t gen = gen_s;
  // workgroup b serves rank (rank + 1 + b) % world: neighbours first, this rank's own row last (its workgroup is the one that waits)
  const int p = (a.rank + 1 + static_cast<int>(blockIdx.x)) % a.world;
  for (int j = 0; j < a.n_layers; ++j) {
    const int n1 = a.n_loc[j] / a.per;                    // columns of one slab run
    const int64_t slab_full = static_cast<int64_t>(n1) * a.world;
    const int64_t row_full = static_cast<int64_t>(a.n_loc[j]) * a.world;   // a full row of the layer
    for (int s = 0; s < a.per; ++s) {
      // row m of the slab run: n1 columns of y_loc[m] -> columns s N / per + rank n1 of the peer's row m (strided slab writes for M > 1: the rows of a
      // batch land in the reference's column order too, nothing is permuted afterwards)
      const uint16_t* src0 = a.y_loc[j] + static_cast<int64_t>(s) * n1;
      uint16_t* dst0 = a.full[p][j] + s * slab_full + static_cast<int64_t>(a.rank) * n1;
      const bool vec = (n1 & 7) == 0 && (a.n_loc[j] & 7) == 0 && ((reinterpret_cast<uintptr_t>(src0) | reinterpret_cast<uintptr_t>(dst0)) & 15) == 0;
      if (vec) {
        const int nv = n1 / 8;
        for (int q = tid; q < a.M * nv; q += 256) {
          const int m = q / nv, i = q - m * nv;
          reinterpret_cast<u32x4*>(dst0 + m * row_full)[i] = reinterpret_cast<const u32x4*>(src0 + static_cast<int64_t>(m) * a.n_loc[j])[i];
        }
      } else {
        for (int q = tid; q < a.M * n1; q += 256) {
          const int m = q / n1, i = q - m * n1;
          dst0[m * row_full + i] = src0[static_cast<int64_t>(m) * a.n_loc[j] + i];
        }
      }
    }
  }
  __threadfence_system();   // this thread's stores are visible at system scope ...
  __syncthreads();          // ... for every thread of the workgroup, before the flag goes up
  if (tid == 0) __hip_atomic_store(a.flags[p] + a.rank, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (p != a.rank) return;
  if (tid < a.world) {
    const uint32_t limit = a.spin_limit ? a.spin_limit : (1u << 22);
    uint32_t n = 0;
    // (generations only grow; the signed difference tolerates the counter's wrap)
    while (static_cast<int32_t>(__hip_atomic_load(a.flags[p] + tid, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - gen) < 0) {
      if (++n >= limit) { __hip_atomic_store(a.status, 1u + static_cast<uint32_t>(tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
}

}  // namespace hqq

extern "C" int hqq_hip_exchange(int n_layers, const void* const* y_loc, const int64_t* N_loc, int64_t M, int nbits, int dtype, int world, int rank,
                                void* const* full, void* const* flags, void* status, uint32_t spin_limit, void* stream) {
  using namespace hqq;
  clear_stale_error();
  if (n_layers < 1 || n_layers > XG_MAXL || world < 1 || world > XG_MAXP || rank < 0 || rank >= world) {
    set_error("hqq_hip_exchange: 1..%d layers and 1..%d ranks per exchange point (got %d layers, rank %d of %d)", XG_MAXL, XG_MAXP, n_layers, rank, world);
    return HQQ_ERR_SHAPE;
  }
  if (M < 1 || M > HQQ_EXCHANGE_MAX_ROWS) { set_error("hqq_hip_exchange: 1..%d activation rows per exchange (got %lld)", HQQ_EXCHANGE_MAX_ROWS, static_cast<long long>(M)); return HQQ_ERR_SHAPE; }
  if (dtype != HQQ_F16 && dtype != HQQ_BF16) { set_error("hqq_hip_exchange: 2-byte activations only (fp16 / bf16)"); return HQQ_ERR_DTYPE; }
  if (nbits != 8 && nbits != 4 && nbits != 3 && nbits != 2 && nbits != 1) { set_error("hqq_hip_exchange: nbits %d", nbits); return HQQ_ERR_NBITS; }
  if (!y_loc || !N_loc || !full || !flags || !status) { set_error("hqq_hip_exchange: null argument"); return HQQ_ERR_SHAPE; }
  XgArgs a;
  a.per = nbits == 3 ? 1 : 8 / nbits;   // 3-bit shards are re-packed from whole rows (hqq_amd/shard.py): one run per rank
  for (int j = 0; j < XG_MAXL; ++j) {
    const int jj = j < n_layers ? j : n_layers - 1;
    if (N_loc[jj] < a.per || N_loc[jj] % a.per != 0 || N_loc[jj] > INT32_MAX / XG_MAXP) { set_error("hqq_hip_exchange: N_loc[%d] = %lld does not split into %d slab runs", jj, static_cast<long long>(N_loc[jj]), a.per); return HQQ_ERR_SHAPE; }
    a.y_loc[j] = static_cast<const uint16_t*>(y_loc[jj]);
    a.n_loc[j] = static_cast<int>(N_loc[jj]);
    if (!y_loc[jj]) { set_error("hqq_hip_exchange: null y_loc[%d]", jj); return HQQ_ERR_SHAPE; }
  }
  for (int p = 0; p < XG_MAXP; ++p) {
    const int pp = p < world ? p : world - 1;
    a.flags[p] = static_cast<uint32_t*>(flags[pp]);
    if (!flags[pp]) { set_error("hqq_hip_exchange: null flag block of rank %d", pp); return HQQ_ERR_SHAPE; }
    for (int j = 0; j < XG_MAXL; ++j) {
      const int jj = j < n_layers ? j : n_layers - 1;
      a.full[p][j] = static_cast<uint16_t*>(full[pp * n_layers + jj]);
      if (!a.full[p][j]) { set_error("hqq_hip_exchange: null row of rank %d, layer %d", pp, jj); return HQQ_ERR_SHAPE; }
    }
  }
  a.status = static_cast<uint32_t*>(status);
  a.n_layers = n_layers; a.world = world; a.rank = rank; a.spin_limit = spin_limit; a.M = static_cast<int>(M);
  hipLaunchKernelGGL(exchange_kernel, dim3(static_cast<unsigned>(world)), dim3(256), 0, as_stream(stream), a);
  return check_launch("hqq_hip_exchange");
}
