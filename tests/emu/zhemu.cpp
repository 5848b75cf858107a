// tests/emu/zhemu.cpp -- host-side wave emulator for kernel LOGIC debugging (test infrastructure only).
// Each of the 64 lanes of a workgroup is a ucontext fiber; a collective is a rendezvous: a lane that arrives
// yields round-robin until all live lanes of the wave have arrived. Single OS thread per grid => deterministic.
#include <ucontext.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <vector>

namespace zhemu {
thread_local uint32_t lane, block, nblocks;
thread_local uint64_t slot[64];
thread_local uint64_t result;
typedef void (*lane_fn)(void*);

static thread_local ucontext_t mainCtx;
static thread_local ucontext_t laneCtx[64];
static thread_local bool alive[64];
static thread_local uint64_t arrivedGen[64];
static thread_local uint64_t gen;
static thread_local int nAlive, nArrived;
static thread_local lane_fn curFn;
static thread_local void* curArg;
static const size_t STACK = 256 * 1024;

static void switch_to_next(uint32_t from)
{
    for (uint32_t k = 1; k <= 64; k++) {
        uint32_t n = (from + k) & 63;
        if (alive[n] && n != from) { lane = n; swapcontext(&laneCtx[from], &laneCtx[n]); lane = from; return; }
    }
}

void collective_wait()
{
    uint32_t me = lane;
    uint64_t g = gen;
    arrivedGen[me] = g + 1;
    nArrived++;
    if (nArrived == nAlive) { nArrived = 0; gen = g + 1; return; }   // last arriver releases everyone
    while (gen == g) {
        switch_to_next(me);
        lane = me;
    }
}

static void trampoline()
{
    uint32_t me = lane;
    curFn(curArg);
    alive[me] = false;
    nAlive--;
    // a lane that exits while others wait in a collective would deadlock real hardware only if the collective
    // needed it; kernels here always exit uniformly. Hand control to another live lane or back to main.
    if (nAlive > 0 && nArrived == nAlive) { nArrived = 0; gen++; }
    for (uint32_t k = 1; k <= 64; k++) {
        uint32_t n = (me + k) & 63;
        if (alive[n]) { lane = n; setcontext(&laneCtx[n]); }
    }
    setcontext(&mainCtx);
}

void run_grid(uint32_t nBlocks, lane_fn fn, void* arg)
{
    std::vector<char*> stacks(64);
    for (int i = 0; i < 64; i++) stacks[i] = (char*)malloc(STACK);
    nblocks = nBlocks; curFn = fn; curArg = arg;
    for (uint32_t b = 0; b < nBlocks; b++) {
        block = b; gen = 0; nArrived = 0; nAlive = 64;
        for (int i = 0; i < 64; i++) {
            getcontext(&laneCtx[i]);
            laneCtx[i].uc_stack.ss_sp = stacks[i];
            laneCtx[i].uc_stack.ss_size = STACK;
            laneCtx[i].uc_link = &mainCtx;
            makecontext(&laneCtx[i], (void (*)())trampoline, 0);
            alive[i] = true; arrivedGen[i] = 0;
        }
        lane = 0;
        swapcontext(&mainCtx, &laneCtx[0]);
        if (nAlive != 0) { fprintf(stderr, "zhemu: block %u ended with %d lanes stuck in a collective\n", b, nAlive); abort(); }
    }
    for (int i = 0; i < 64; i++) free(stacks[i]);
}
}  // namespace zhemu
