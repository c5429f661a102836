// launch_trace.hpp -- every kernel launch of the library goes through BF_LAUNCH, which notes the kernel's host pointer in the calling
// thread's trace when one is open (bf_trace_begin / bf_trace_end in bfcore.h).  Users: tools/dispatch_table.py (which kernels run for a
// node / period / layout / microphone count), bench.py (its `traffic` figures are printed only beside the kernels they were measured
// on).  Closed trace: one thread_local load and a branch per launch.
#pragma once

#include <hip/hip_runtime.h>

namespace bf {
void trace_note(const void *host_fn);  // capi.cpp
}

#define BF_LAUNCH(kern, ...)                                           \
    do {                                                               \
        ::bf::trace_note(reinterpret_cast<const void *>(kern));       \
        hipLaunchKernelGGL(kern, __VA_ARGS__);                         \
    } while (0)
