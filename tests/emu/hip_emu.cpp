// TEST INFRASTRUCTURE ONLY - see hip_emu.h.
#include "hip_emu.h"

namespace sgx_emu {

BlockState* g_block = nullptr;
thread_local uint3_emu t_threadIdx, t_blockIdx;
thread_local dim3 t_blockDim, t_gridDim;
thread_local int t_lane, t_wave;
thread_local unsigned t_xgen;

static std::mutex g_launch_mutex;

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    std::lock_guard<std::mutex> lk(g_launch_mutex);
    const int nthreads = (int)(block.x * block.y * block.z);
    const long nblocks = (long)grid.x * grid.y * grid.z;
    if (nthreads <= 0 || nblocks <= 0) return;
    if (smem > sizeof(BlockState::dyn_smem)) {
        fprintf(stderr, "sgx_emu: dynamic smem %zu too large\n", smem);
        abort();
    }
    static BlockState* bs = new BlockState();
    g_block = bs;
    const int nwaves = (nthreads + 63) / 64;
    if ((int)bs->waves.size() < nwaves) {
        std::vector<WaveState> nv(nwaves);
        bs->waves.swap(nv);
    }
    // SGX_EMU_SHUFFLE=<seed>: run the workgroups in a pseudo-random order (kernels that hand work between workgroups - arrival tickets -
    // must not depend on the dispatch order)
    std::vector<long> order(nblocks);
    for (long b = 0; b < nblocks; ++b) order[b] = b;
    if (const char* sh = getenv("SGX_EMU_SHUFFLE")) {
        unsigned long long st = strtoull(sh, nullptr, 10) * 6364136223846793005ull + 1442695040888963407ull;
        for (long b = nblocks - 1; b > 0; --b) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(order[b], order[(long)((st >> 33) % (unsigned long long)(b + 1))]);
        }
    }
    Barrier outer;
    outer.reset(nthreads);
    auto worker = [&](int tid) {
        t_blockDim = block;
        t_gridDim = grid;
        t_threadIdx.x = tid % block.x;
        t_threadIdx.y = (tid / block.x) % block.y;
        t_threadIdx.z = tid / (block.x * block.y);
        t_lane = tid & 63;
        t_wave = tid >> 6;
        for (long bi = 0; bi < nblocks; ++bi) {
            const long b = order[bi];
            if (tid == 0) {
                bs->bar.reset(nthreads);
                for (int w = 0; w < nwaves; ++w) {
                    int n = nthreads - w * 64;
                    bs->waves[w].bar.reset(n > 64 ? 64 : n);
                }
            }
            outer.wait();
            t_blockIdx.x = (unsigned)(b % grid.x);
            t_blockIdx.y = (unsigned)((b / grid.x) % grid.y);
            t_blockIdx.z = (unsigned)(b / ((long)grid.x * grid.y));
            t_xgen = 0;
            body();
            bs->bar.drop();
            bs->waves[t_wave].bar.drop();
            outer.wait();
        }
    };
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (int t = 0; t < nthreads; ++t) th.emplace_back(worker, t);
    for (auto& t : th) t.join();
}

}  // namespace sgx_emu
