// Stand-in for the reference's src/core/cpu_profiling.h (Win32 QueryPerformanceCounter profiler): profiling blocks
// compile to nothing.  Test infrastructure (oracle/refbuild).
#pragma once
#define CPU_PROFILE_BLOCK(name) ((void)0)
#define CPU_PRINT_PROFILE_BLOCK(name) ((void)0)
#define CPU_PROFILE_STAT(label, value) ((void)0)
