// tools/hipsim/hipsim.cpp -- scheduler of the CPU SIMT emulator (see include/hip/hip_runtime.h).
// TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>

dim3 threadIdx, blockIdx, blockDim, gridDim;

// Minimal x86-64 SysV context switch: push callee-saved registers, swap stack pointers, pop, return.
// (glibc's swapcontext makes a sigprocmask system call per switch, ~50x slower.)
asm(R"(
.text
.globl hipsim_switch
.type hipsim_switch,@function
hipsim_switch:
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
.size hipsim_switch,.-hipsim_switch
)");

namespace hipsim {

static void trampoline() { fiber_main(); fprintf(stderr, "hipsim: fiber resumed after exit\n"); abort(); }

// lays out a fresh stack so that the first hipsim_switch into it "returns" to trampoline()
static void* freshStack(std::vector<char>& stack)
{
  uintptr_t top = ((uintptr_t)stack.data() + stack.size()) & ~(uintptr_t)15;
  void** sp = (void**)top;
  *--sp = nullptr;                    // fake return address of trampoline (keeps the ABI's 16-byte alignment)
  *--sp = (void*)&trampoline;         // popped by `ret`
  for (int i = 0; i < 6; i++) *--sp = nullptr;    // rbp, rbx, r12..r15
  return (void*)sp;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body)
{
  State& s = S();
  const int nThreads = (int)(block.x * block.y * block.z);
  if (nThreads <= 0 || nThreads > 1024) { fprintf(stderr, "hipsim: bad block size %d\n", nThreads); abort(); }
  if (s.cur >= 0) { fprintf(stderr, "hipsim: nested launch\n"); abort(); }
  const size_t kStack = 256 * 1024;
  if ((int)s.fibers.size() < nThreads) s.fibers.resize(nThreads);
  for (int t = 0; t < nThreads; t++)
    if (s.fibers[t].stack.size() != kStack) s.fibers[t].stack.resize(kStack);
  s.body = body;
  s.grid = grid; s.blk = block;
  gridDim = grid; blockDim = block;
  const int nWaves = (nThreads + 63) / 64;

  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++)
      {
        blockIdx = dim3(bx, by, bz);
        s.block = Group(); s.block.live = nThreads;
        s.waves.assign(nWaves, Group());
        memset(s.stamp, 0, sizeof(s.stamp));
        for (int t = 0; t < nThreads; t++)
        {
          Fiber& f = s.fibers[t];
          f.done = false;
          f.lin = t;
          f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          s.waves[t / 64].live++;
          f.collGen = 0;
          f.sp = freshStack(f.stack);
        }
        int remaining = nThreads;
        long idleRounds = 0;
        while (remaining > 0)
        {
          unsigned sig = s.block.gen;
          for (const Group& w : s.waves) sig = sig * 31 + w.gen;
          int before = remaining;
          for (int t = 0; t < nThreads; t++)
          {
            Fiber& f = s.fibers[t];
            if (f.done) continue;
            s.cur = t;
            threadIdx = f.tid;
            hipsim_switch(&s.schedSp, f.sp);
            if (f.done) remaining--;
          }
          unsigned sig2 = s.block.gen;
          for (const Group& w : s.waves) sig2 = sig2 * 31 + w.gen;
          if (sig2 == sig && before == remaining)
          {
            if (++idleRounds > 100000)
            {
              fprintf(stderr, "hipsim: deadlock in block (%u,%u,%u): %d fibers stuck (divergent collective or a spin on a later block?)\n",
                bx, by, bz, remaining);
              abort();
            }
          }
          else idleRounds = 0;
        }
        s.cur = -1;
      }
}

}    // namespace hipsim
