// Fiber scheduler for the HIP-on-CPU emulator (test infrastructure, see hip/hip_runtime.h).
// One OS thread runs one block at a time; every GPU thread of the block is a fiber with its
// own stack, switched cooperatively at __syncthreads() / wave collectives.
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <chrono>
#include <thread>
#include <vector>

thread_local emu_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

namespace {

constexpr size_t kStack = 96 * 1024;

struct Fiber {
    void* sp = nullptr;
    unsigned tid = 0;
    bool done = false;
    int wait_kind = 0;  // 0 runnable, 1 block barrier, 2 wave barrier
    unsigned wait_gen = 0;
};

struct Worker {
    std::vector<char> stacks;
    std::vector<Fiber> fibers;
    void* sched_sp = nullptr;
    int cur = -1;
    unsigned nthreads = 0;
    unsigned block_gen = 0, block_cnt = 0, live = 0;
    std::vector<unsigned> wave_gen, wave_cnt, wave_live, wave_phase;
    std::vector<float> wave_scratch;
    const std::function<void()>* body = nullptr;
};

thread_local Worker* W = nullptr;

void fiber_main() {
    Worker* w = W;
    (*w->body)();
    Fiber& f = w->fibers[w->cur];
    f.done = true;
    w->live--;
    w->wave_live[f.tid / 64]--;
    // a finished thread no longer takes part in barriers: release peers if it was the last
    if (w->block_cnt && w->block_cnt == w->live) { w->block_cnt = 0; w->block_gen++; }
    unsigned wv = f.tid / 64;
    if (w->wave_cnt[wv] && w->wave_cnt[wv] == w->wave_live[wv]) { w->wave_cnt[wv] = 0; w->wave_gen[wv]++; w->wave_phase[wv] ^= 1; }
    emu_switch(&f.sp, w->sched_sp);
    abort();
}

void run_block(Worker& w, const std::function<void()>& body, dim3 block, emu_uint3 bidx, dim3 grid) {
    const unsigned n = block.x * block.y * block.z;
    if (w.nthreads < n) {
        w.stacks.assign((size_t)n * kStack, 0);
        w.fibers.assign(n, Fiber());
        w.nthreads = n;
    }
    const unsigned nw = (n + 63) / 64;
    w.wave_gen.assign(nw, 0); w.wave_cnt.assign(nw, 0); w.wave_live.assign(nw, 0); w.wave_phase.assign(nw, 0);
    w.wave_scratch.assign((size_t)nw * 256, 0.f);
    w.block_gen = 0; w.block_cnt = 0; w.live = n; w.body = &body;
    for (unsigned t = 0; t < n; ++t) {
        Fiber& f = w.fibers[t];
        f.tid = t; f.done = false; f.wait_kind = 0; f.wait_gen = 0;
        char* top = w.stacks.data() + (size_t)(t + 1) * kStack;
        top = (char*)((uintptr_t)top & ~(uintptr_t)15);
        void** sp = (void**)(top - 16);
        sp[0] = (void*)&fiber_main;   // return address, 16-byte aligned slot
        sp -= 6;                       // rbp rbx r12 r13 r14 r15
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;
        f.sp = sp;
        w.wave_live[t / 64]++;
    }
    blockIdx = bidx; blockDim = block; gridDim = grid;
    unsigned remaining = n;
    while (remaining) {
        bool progressed = false;
        for (unsigned t = 0; t < n; ++t) {
            Fiber& f = w.fibers[t];
            if (f.done) continue;
            if (f.wait_kind == 1 && f.wait_gen == w.block_gen) continue;
            if (f.wait_kind == 2 && f.wait_gen == w.wave_gen[t / 64]) continue;
            f.wait_kind = 0;
            w.cur = (int)t;
            threadIdx.x = t % block.x;
            threadIdx.y = (t / block.x) % block.y;
            threadIdx.z = t / (block.x * block.y);
            emu_switch(&w.sched_sp, f.sp);
            progressed = true;
            if (f.done) remaining--;
        }
        if (!progressed) { fprintf(stderr, "emu: deadlock (divergent barrier?)\n"); abort(); }
    }
}

}  // namespace

void emu_sync_block() {
    Worker* w = W;
    Fiber& f = w->fibers[w->cur];
    if (++w->block_cnt == w->live) { w->block_cnt = 0; w->block_gen++; return; }
    f.wait_kind = 1; f.wait_gen = w->block_gen;
    emu_switch(&f.sp, w->sched_sp);
}

void emu_sync_wave() {
    Worker* w = W;
    Fiber& f = w->fibers[w->cur];
    const unsigned wv = f.tid / 64;
    if (++w->wave_cnt[wv] == w->wave_live[wv]) { w->wave_cnt[wv] = 0; w->wave_gen[wv]++; w->wave_phase[wv] ^= 1; return; }
    f.wait_kind = 2; f.wait_gen = w->wave_gen[wv];
    emu_switch(&f.sp, w->sched_sp);
}

// MFMA operand exchange in ONE runtime call (the emulated conv kernels spend their time here): lane l stores its a / b
// operand, waits for the wave, returns the buffer of the phase it wrote in.
float* emu_wave_exchange2(float a, float b) {
    Worker* w = W;
    Fiber& f = w->fibers[w->cur];
    const unsigned wv = f.tid / 64, l = f.tid & 63;
    float* s = w->wave_scratch.data() + (size_t)wv * 256 + w->wave_phase[wv] * 128;
    s[l] = a;
    s[64 + l] = b;
    if (++w->wave_cnt[wv] == w->wave_live[wv]) { w->wave_cnt[wv] = 0; w->wave_gen[wv]++; w->wave_phase[wv] ^= 1; return s; }
    f.wait_kind = 2; f.wait_gen = w->wave_gen[wv];
    emu_switch(&f.sp, w->sched_sp);
    return s;
}

// NOTE: a lane reads the exchange buffer of the phase it wrote in; the phase flips when the
// last lane arrives, so readers must latch the pointer BEFORE the sync (intrin.h does).
float* emu_wave_scratch() { return W->wave_scratch.data() + (size_t)(W->fibers[W->cur].tid / 64) * 256; }
int emu_wave_phase() { return (int)W->wave_phase[W->fibers[W->cur].tid / 64]; }

// A spinning consumer block gives its core away; on a loaded host (8 cores shared with a compiler) plain yields can burn
// through the kernel's spin limit before the producer's thread is scheduled at all, so every 256th call sleeps.
void emu_yield_os() {
    static thread_local unsigned calls = 0;
    if ((++calls & 255u) == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
    else std::this_thread::yield();
}

void emu_launch(std::function<void()> body, dim3 grid, dim3 block) {
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    unsigned nthr = std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), nblocks);
    if (const char* e = getenv("CLSLAM_EMU_THREADS")) nthr = std::max(1, atoi(e));
    nthr = std::min<size_t>(nthr, nblocks);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        static thread_local Worker worker;
        W = &worker;
        for (;;) {
            size_t b = next.fetch_add(1);
            if (b >= nblocks) break;
            emu_uint3 bi{(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y))};
            run_block(worker, body, block, bi, grid);
        }
    };
    if (nthr == 1) { work(); return; }
    std::vector<std::thread> ts;
    for (unsigned i = 0; i < nthr; ++i) ts.emplace_back(work);
    for (auto& t : ts) t.join();
}
