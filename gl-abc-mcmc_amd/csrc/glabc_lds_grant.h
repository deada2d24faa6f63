// glabc_lds_grant.h -- host side: dynamic LDS beyond the 48 KiB default has to be granted per kernel AND per device
// (hipFuncSetAttribute acts on the current device's copy of the function).  The launchers remember what they were granted
// per device, so a process that drives several GPUs (one stream each) gets the attribute set on every one of them.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

namespace glabc {

struct LdsGrant {
    size_t bytes[64] = {};                  // by device ordinal (mod 64); 0 = nothing granted yet
};

// returns false if the runtime refuses; `floor` = what needs no grant (48 KiB)
inline bool grant_dynamic_lds(LdsGrant& g, const void* fn, size_t bytes, size_t floor = 48 * 1024)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    size_t& have = g.bytes[(unsigned)dev & 63u];
    if (bytes <= floor || bytes <= have) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    have = bytes;
    return true;
}

}  // namespace glabc
