// ptx_shard_samples lives in ptx_multi.hip (device TU); restate for the host-only sanitizer build
#include "../include/ptx.h"
extern "C" int ptx_shard_samples(uint32_t first, uint32_t count, uint32_t n, uint32_t g, uint32_t* f, uint32_t* c) { unsigned long long lo=(unsigned long long)count*g/n, hi=(unsigned long long)count*(g+1)/n; *f=first+lo; *c=hi-lo; return 0; }
