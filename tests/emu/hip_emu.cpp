// Fiber scheduler for tests/emu/hip_emu.h.  TEST INFRASTRUCTURE ONLY (see header).
#include "hip_emu.h"

#include <mutex>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

namespace emu {

thread_local Block* g_blk = nullptr;
thread_local Lane* g_lane = nullptr;

void yield_lane() {
    Lane* me = g_lane;
    swapcontext(&me->ctx, &g_blk->main_ctx);
}

static void trampoline() {
    (*g_blk->body)();
    g_lane->done = true;
    // returning follows uc_link back to the scheduler
}

static void run_block(Block& b, std::vector<char*>& stacks) {
    const unsigned nthreads = b.bdim.x;
    const unsigned nwaves = (nthreads + WAVE - 1) / WAVE;
    b.lanes.assign(nthreads, Lane());
    b.waves.assign(nwaves, WaveState());
    b.blk_active = nthreads; b.blk_arrived = 0; b.blk_gen = 0;
    while (stacks.size() < nthreads) stacks.push_back((char*)malloc(STACK_BYTES));
    for (unsigned t = 0; t < nthreads; ++t) {
        Lane& l = b.lanes[t];
        l.tid = t; l.stack = stacks[t];
        b.waves[t / WAVE].active++;
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack;
        l.ctx.uc_stack.ss_size = STACK_BYTES;
        l.ctx.uc_link = &b.main_ctx;
        makecontext(&l.ctx, (void (*)())trampoline, 0);
    }
    g_blk = &b;
    unsigned remaining = nthreads;
    while (remaining) {
        bool progress = false;
        for (unsigned t = 0; t < nthreads; ++t) {
            Lane& l = b.lanes[t];
            if (l.done) continue;
            if (l.wait_kind == 1 && b.waves[t / WAVE].gen == l.wait_gen) continue;
            if (l.wait_kind == 2 && b.blk_gen == l.wait_gen) continue;
            l.wait_kind = 0;
            g_lane = &l;
            swapcontext(&b.main_ctx, &l.ctx);
            progress = true;
            if (l.done) {
                --remaining;
                WaveState& w = b.waves[t / WAVE];
                if (--w.active > 0 && w.arrived >= w.active) { w.arrived = 0; w.gen++; }
                if (--b.blk_active > 0 && b.blk_arrived >= b.blk_active) { b.blk_arrived = 0; b.blk_gen++; }
            }
        }
        if (!progress) {
            fprintf(stderr, "hip_emu: deadlock in block (%u,%u): %u lanes stuck (divergent collective?)\n",
                    b.bidx.x, b.bidx.y, remaining);
            abort();
        }
    }
    g_blk = nullptr; g_lane = nullptr;
}

static void watchdog(int) {
    void* bt[64];
    int n = backtrace(bt, 64);
    fprintf(stderr, "hip_emu watchdog: kernel still running; backtrace of the active fiber:\n");
    backtrace_symbols_fd(bt, n, 2);
    _exit(3);
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    if (const char* w = getenv("NEURAY_EMU_WATCHDOG")) { signal(SIGALRM, watchdog); alarm(atoi(w)); }
    const unsigned nblocks = grid.x * grid.y * grid.z;
    unsigned nthr = std::thread::hardware_concurrency();
    const char* env = getenv("NEURAY_EMU_THREADS");
    if (env) nthr = (unsigned)atoi(env);
    if (nthr < 1) nthr = 1;
    if (nthr > nblocks) nthr = nblocks;
    std::atomic<unsigned> next(0);
    auto worker = [&]() {
        std::vector<char*> stacks;
        std::vector<char> dyn(smem + 64);
        for (;;) {
            unsigned i = next.fetch_add(1);
            if (i >= nblocks) break;
            Block b;
            b.bidx = dim3(i % grid.x, (i / grid.x) % grid.y, i / (grid.x * grid.y));
            b.bdim = block; b.gdim = grid; b.body = &body;
            b.dyn_smem = (char*)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
            run_block(b, stacks);
        }
        for (char* s : stacks) free(s);
    };
    if (nthr == 1) { worker(); alarm(0); return; }
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < nthr; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
    alarm(0);
}

}  // namespace emu
