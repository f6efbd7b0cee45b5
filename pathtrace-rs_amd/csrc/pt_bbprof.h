// pt_bbprof.h -- development aid (tools/bbprof.py): execution counts of the kernels' basic blocks, bumped by scalar atomics that
// the tool writes into ONE unit's ASSEMBLY (the unit compiled with -DPT_BBPROF); pt_bbprof_dump hands them to the tool.
#pragma once
#include <cstdio>
#include <hip/hip_runtime.h>
extern "C" {
__device__ __attribute__((used, visibility("default"))) unsigned long long pt_bbprof[65536];   // [0, 32768) executions of block i; [32768, 65536) lanes that were switched on, summed over them
}
extern "C" __attribute__((visibility("default"), used)) inline int pt_bbprof_dump(const char *path) {
    static unsigned long long host[65536];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(host, HIP_SYMBOL(pt_bbprof), sizeof host) != hipSuccess) return -1;
    FILE *f = fopen(path, "w");
    if (!f) return -2;
    for (int i = 0; i < 65536; ++i)
        if (host[i]) fprintf(f, "%d %llu\n", i, host[i]);
    fclose(f);
    return 0;
}
