// TEST INFRASTRUCTURE ONLY - see hip_emu.h.
#include "hip_emu.h"

#include <sys/mman.h>

namespace sgx_emu {

BlockState* g_block = nullptr;
thread_local uint3_emu t_threadIdx, t_blockIdx;
thread_local dim3 t_blockDim, t_gridDim;
thread_local int t_lane, t_wave;
thread_local unsigned t_xgen;

static std::mutex g_launch_mutex;

// ---- fibers ----------------------------------------------------------------------------------------------------------------------
// A context switch saves the callee-saved registers on the current stack, stores the stack pointer, loads the other one and pops its
// registers (x86-64 System V; no signal mask, no syscall).  Other targets fall back to swapcontext.
#if defined(__x86_64__)
extern "C" void sgx_emu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl sgx_emu_switch
    .type sgx_emu_switch,@function
sgx_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size sgx_emu_switch,.-sgx_emu_switch
)");
#else
#include <ucontext.h>
#endif

struct Fiber {
#if defined(__x86_64__)
    void* sp = nullptr;
#else
    ucontext_t ctx;
#endif
    char* stack = nullptr;
    bool done = true;
    uint3_emu tidx{0, 0, 0};
    int lane = 0, wave = 0;
    unsigned xgen = 0;
    const unsigned long* wait_gen = nullptr;  // blocked in a barrier while *wait_gen == wait_seen
    unsigned long wait_seen = 0;
};
static constexpr size_t kStack = 512 * 1024;  // per fiber; mapped lazily (kernels keep accumulator tiles and staging registers as locals)
static std::vector<Fiber> g_fibers;
static int g_cur = -1;
static const std::function<void()>* g_body = nullptr;
#if defined(__x86_64__)
static void* g_sched_sp = nullptr;
#else
static ucontext_t g_sched_ctx;
#endif

static void to_scheduler() {
    Fiber& f = g_fibers[g_cur];
    f.xgen = t_xgen;
#if defined(__x86_64__)
    sgx_emu_switch(&f.sp, g_sched_sp);
#else
    swapcontext(&f.ctx, &g_sched_ctx);
#endif
}
void fiber_wait(const unsigned long* gen, unsigned long seen) {
    Fiber& f = g_fibers[g_cur];
    f.wait_gen = gen;
    f.wait_seen = seen;
    do {
        to_scheduler();
    } while (*gen == seen);
    g_fibers[g_cur].wait_gen = nullptr;
}
static void fiber_main() {
    (*g_body)();
    Fiber& f = g_fibers[g_cur];
    g_block->bar.drop();
    g_block->waves[f.wave].bar.drop();
    f.done = true;
    to_scheduler();
    abort();  // a finished fiber is never resumed
}
static void fiber_prepare(Fiber& f) {
    if (!f.stack) {
        f.stack = static_cast<char*>(mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
        if (f.stack == MAP_FAILED) {
            perror("sgx_emu: fiber stack");
            abort();
        }
        mprotect(f.stack, 4096, PROT_NONE);  // guard page: a kernel that outgrows its stack faults instead of writing into its neighbour's
    }
    f.done = false;
    f.wait_gen = nullptr;
    f.xgen = 0;
#if defined(__x86_64__)
    // initial frame: what sgx_emu_switch pops - control words, six registers - then fiber_main as the return address; above it one slot
    // so that fiber_main starts with the stack alignment of a called function (rsp = 16 k + 8)
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15);
    uint64_t* sp = reinterpret_cast<uint64_t*>(top);
    *--sp = 0;                                           // (alignment slot / fake return address of fiber_main)
    *--sp = reinterpret_cast<uint64_t>(&fiber_main);     // ret target
    for (int i = 0; i < 6; ++i) *--sp = 0;               // rbp rbx r12 r13 r14 r15
    uint32_t cw[2];
    asm volatile("stmxcsr %0" : "=m"(cw[0]));
    uint16_t fcw;
    asm volatile("fnstcw %0" : "=m"(fcw));
    cw[1] = fcw;
    uint64_t word;
    memcpy(&word, cw, 8);
    *--sp = word;                                        // mxcsr | x87 control word
    f.sp = sp;
#else
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, fiber_main, 0);
#endif
}
static void fiber_resume(int i) {
    Fiber& f = g_fibers[i];
    g_cur = i;
    t_threadIdx = f.tidx;
    t_lane = f.lane;
    t_wave = f.wave;
    t_xgen = f.xgen;
#if defined(__x86_64__)
    sgx_emu_switch(&g_sched_sp, f.sp);
#else
    swapcontext(&g_sched_ctx, &f.ctx);
#endif
    g_cur = -1;
}

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
    if ((int)g_fibers.size() < nthreads) g_fibers.resize(nthreads);
    g_body = &body;
    t_blockDim = block;
    t_gridDim = grid;
    for (long bi = 0; bi < nblocks; ++bi) {
        const long b = order[bi];
        bs->bar.reset(nthreads);
        for (int w = 0; w < nwaves; ++w) {
            int n = nthreads - w * 64;
            bs->waves[w].bar.reset(n > 64 ? 64 : n);
        }
        t_blockIdx.x = (unsigned)(b % grid.x);
        t_blockIdx.y = (unsigned)((b / grid.x) % grid.y);
        t_blockIdx.z = (unsigned)(b / ((long)grid.x * grid.y));
        for (int tid = 0; tid < nthreads; ++tid) {
            Fiber& f = g_fibers[tid];
            f.tidx.x = tid % block.x;
            f.tidx.y = (tid / block.x) % block.y;
            f.tidx.z = tid / (block.x * block.y);
            f.lane = tid & 63;
            f.wave = tid >> 6;
            fiber_prepare(f);
        }
        // round robin: resume every fiber that is not waiting for a barrier whose generation has not moved
        int alive = nthreads;
        while (alive) {
            bool progressed = false;
            for (int tid = 0; tid < nthreads; ++tid) {
                Fiber& f = g_fibers[tid];
                if (f.done || (f.wait_gen && *f.wait_gen == f.wait_seen)) continue;
                progressed = true;
                fiber_resume(tid);
                if (f.done) --alive;
            }
            if (!progressed && alive) {
                fprintf(stderr, "sgx_emu: deadlock - %d lane(s) of workgroup %ld wait at barriers the others never reach\n", alive, b);
                abort();
            }
        }
    }
    g_body = nullptr;
}

}  // namespace sgx_emu
