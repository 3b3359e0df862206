// Minimal x86-64 SysV context switch for the emulator's work-item fibers (tests/emu/shim/hip/hip_runtime.h):
//   void hipemu_switch(void** save_sp, void* load_sp)
// pushes the callee-saved registers, stores the stack pointer through save_sp, adopts load_sp and pops the same
// registers.  glibc's swapcontext does this plus a sigprocmask system call per switch, which dominated run time.
    .text
    .globl hipemu_switch
    .type hipemu_switch, @function
hipemu_switch:
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
    .size hipemu_switch, .-hipemu_switch
    .section .note.GNU-stack,"",@progbits
