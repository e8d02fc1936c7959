// Host-side half of the multi-GPU DSGD ring (SURVEY.md 8(b) `mfsgd_dsgd_*`, 8(e)): who owns which users, which item
// shard a rank holds in which sub-epoch, whom it sends to, and which block grid to use.  Pure integer logic, no device
// is touched; the compute step of a rank is mfsgd_hip_epoch_* on the grid of the shard it holds, the exchange is the
// host's RCCL send/recv of Q[item_edges[q] .. item_edges[q+1]) and b_i (recommender_amd/dsgd.py does exactly this over
// torch.distributed).  The reference has no multi-device path; the exclusivity rule is its FPSGD invariant
// (RayBlockScheduler.py:50-62) one level up.
#include <algorithm>
#include <cstdint>

#include "common.h"

extern "C" {

int mfsgd_dsgd_plan(int64_t n_users, int64_t n_items, int32_t world, int32_t ubp, int32_t ibp, int64_t *user_edges,
                    int64_t *item_edges, int32_t *Gr_out, int32_t *Gc_out)
{
    MF_ARG(n_users > 0 && n_items > 0 && world > 0 && ubp > 0 && ibp > 0, "dsgd_plan: bad shape argument");
    MF_ARG(user_edges && item_edges, "dsgd_plan: null output");
    const int64_t Gr = (n_users + ubp - 1) / ubp, Gc = (n_items + ibp - 1) / ibp;
    const int64_t rows = (Gr + world - 1) / world, cols = (Gc + world - 1) / world;   // stripes per rank / per shard
    for (int32_t g = 0; g <= world; ++g) {
        user_edges[g] = std::min<int64_t>(n_users, (int64_t)g * rows * ubp);
        item_edges[g] = std::min<int64_t>(n_items, (int64_t)g * cols * ibp);
    }
    user_edges[world] = n_users;
    item_edges[world] = n_items;
    if (Gr_out) *Gr_out = (int32_t)Gr;
    if (Gc_out) *Gc_out = (int32_t)Gc;
    return MFSGD_OK;
}

int mfsgd_dsgd_schedule(int32_t world, int32_t rank, int32_t step, int32_t *shard, int32_t *send_to, int32_t *recv_from)
{
    MF_ARG(world > 0 && rank >= 0 && rank < world && step >= 0, "dsgd_schedule: rank %d / step %d outside a world of %d", rank,
           step, world);
    if (shard) *shard = (rank + step) % world;
    if (send_to) *send_to = (rank + world - 1) % world;
    if (recv_from) *recv_from = (rank + 1) % world;
    return MFSGD_OK;
}

int mfsgd_dsgd_choose_grid(int64_t n_users, int64_t n_items, int64_t n_train, int32_t world, int32_t k, int32_t elem_size,
                           int64_t max_item_degree, int32_t n_cus, int32_t *ubp_out, int32_t *ibp_out)
{
    MF_ARG(n_users > 0 && n_items > 0 && n_train >= 0 && world > 0 && k > 0 && (elem_size == 4 || elem_size == 8) && n_cus > 0,
           "dsgd_choose_grid: bad argument");
    MF_ARG(ubp_out && ibp_out, "dsgd_choose_grid: null output");
    // one rank's sub-epoch kernel, measured (profiles/r2_dsgd_rank_sim.txt): a stripe's tenure is the chain of its most
    // rated item plus a block transition per row; the ratings matter only when the stripes are few
    const double block_us = 15.0, rate = 15.0, link_us[2] = {0.09, 0.22};   // hot-item walker, item tickets (dsgd.CHAIN_LINK_US)
    const int64_t ibp_max = std::max<int64_t>(1, (int64_t)(120 * 1024) / ((int64_t)(k + 1) * elem_size + 4));
    const double per = (double)n_train / ((double)world * world);
    const double top = max_item_degree > 0 ? (double)max_item_degree : 0.0016 * (double)n_train;
    const int64_t cmin = std::max<int64_t>(1, (n_items + world * ibp_max - 1) / (world * ibp_max));
    double best_t = -1;
    int64_t best_rows = 0, best_cols = 0;
    for (int64_t cols = std::max<int64_t>(cmin, 8); cols <= std::min<int64_t>(n_cus, std::max<int64_t>(1, n_items / world)); cols += 4) {
        const int64_t rows = std::min<int64_t>(cols + std::max<int64_t>(1, cols / 8), std::max<int64_t>(1, n_users / world));
        const int64_t ibp = (n_items + cols * world - 1) / (cols * world);
        const bool piped = (k == 64 || k == 128 || k == 256) && elem_size == 4 &&
                           mfsgd_lds_bytes((int32_t)ibp, (int32_t)rows, k, 4, 0) <= 160 * 1024;   // where the walker runs
        const double chain = top / world * link_us[piped ? 0 : 1];
        const double t = std::max(chain + rows * block_us, per / (cols * rate));
        if (best_t < 0 || t < best_t - 1e-9) { best_t = t; best_rows = rows; best_cols = cols; }
    }
    if (best_t < 0) {   // stripes limited by the LDS, more of them than CUs: a workgroup takes several in turn
        best_rows = std::min<int64_t>(n_cus + n_cus / 8, std::max<int64_t>(1, n_users / world));
        best_cols = cmin;
    }
    *ibp_out = (int32_t)std::max<int64_t>(1, std::min<int64_t>(ibp_max, (n_items + best_cols * world - 1) / (best_cols * world)));
    *ubp_out = (int32_t)std::max<int64_t>(1, (n_users + best_rows * world - 1) / (best_rows * world));
    return MFSGD_OK;
}

}  // extern "C"
