// TEST INFRASTRUCTURE ONLY - see hip_emu.h.
#include "hip_emu.h"

namespace sgx_emu {

BlockState* g_block = nullptr;
thread_local uint3_emu t_threadIdx, t_blockIdx;
thread_local dim3 t_blockDim, t_gridDim;
thread_local int t_lane, t_wave;
thread_local unsigned t_xgen;

static std::mutex g_launch_mutex;

// Persistent worker pool: creating 256-1024 OS threads per kernel launch dominated the run time of the whole-model tests (thousands of
// launches).  Workers park on their OWN condition variable (a launch wakes exactly the threads it needs), live for the life of the
// process and are never joined; the pool state is heap-allocated and intentionally leaked so that nothing is destroyed under a parked
// thread at exit.  Launches are serialised by g_launch_mutex, so there is one job at a time.
namespace {
struct PoolWorker {
    std::mutex m;
    std::condition_variable cv;
    unsigned long ticket = 0;  // incremented by launch() to hand this worker the current job
};
struct Pool {
    std::vector<PoolWorker*> workers;
    const std::function<void(int)>* job = nullptr;
    std::mutex done_m;
    std::condition_variable done_cv;
    int remaining = 0;
};
Pool* g_pool = nullptr;

void pool_thread(Pool* pool, PoolWorker* w, int tid) {
    unsigned long seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { return w->ticket != seen; });
            seen = w->ticket;
        }
        (*pool->job)(tid);
        std::lock_guard<std::mutex> lk(pool->done_m);
        if (--pool->remaining == 0) pool->done_cv.notify_one();
    }
}

void run_on_pool(int nthreads, const std::function<void(int)>& fn) {
    if (!g_pool) g_pool = new Pool();
    Pool* pool = g_pool;
    while ((int)pool->workers.size() < nthreads) {
        PoolWorker* w = new PoolWorker();
        const int tid = (int)pool->workers.size();
        pool->workers.push_back(w);
        std::thread(pool_thread, pool, w, tid).detach();
    }
    pool->job = &fn;
    {
        std::lock_guard<std::mutex> lk(pool->done_m);
        pool->remaining = nthreads;
    }
    for (int t = 0; t < nthreads; ++t) {
        PoolWorker* w = pool->workers[t];
        {
            std::lock_guard<std::mutex> lk(w->m);
            ++w->ticket;
        }
        w->cv.notify_one();
    }
    std::unique_lock<std::mutex> lk(pool->done_m);
    pool->done_cv.wait(lk, [&] { return pool->remaining == 0; });
}
}  // namespace

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
    Barrier outer;
    outer.reset(nthreads);
    const std::function<void(int)> worker = [&](int tid) {
        t_blockDim = block;
        t_gridDim = grid;
        t_threadIdx.x = tid % block.x;
        t_threadIdx.y = (tid / block.x) % block.y;
        t_threadIdx.z = tid / (block.x * block.y);
        t_lane = tid & 63;
        t_wave = tid >> 6;
        for (long b = 0; b < nblocks; ++b) {
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
    run_on_pool(nthreads, worker);
}

}  // namespace sgx_emu
