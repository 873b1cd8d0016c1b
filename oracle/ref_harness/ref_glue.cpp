// oracle/ref_harness/ref_glue.cpp -- TEST INFRASTRUCTURE, build container only.
//
// Link-time glue for the PARTIAL reference build.  Core/Utils/Memory.cpp cannot be compiled here (it
// includes <Windows.h> unconditionally, line 8), so the four allocation entry points it would define are
// provided below over posix_memalign/free.  No arithmetic on the hot path depends on them.  They are
// declared (not defined) in the reference's own header Core/Utils/Memory.h:15-29, which is what is
// included here.  Likewise Core/Utils/MemoryHelpers.cpp (needs <intrin.h>) would define LargeMemCopy, which
// Bitmap::Copy references: a plain memcpy here.
#include <stdlib.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "PCH.h"
#include "Utils/Memory.h"
#include "Utils/MemoryHelpers.h"

namespace rt {

void InitMemory(const MemoryInitOptions&) {}

void* DefaultAllocator::Allocate(size_t size, size_t alignment)
{
    void* ptr = nullptr;
    alignment = std::max(alignment, sizeof(void*));
    size_t a = 1; while (a < alignment) a <<= 1;   // posix_memalign wants a power of two
    if (posix_memalign(&ptr, a, size ? size : 1) != 0) return nullptr;
    return ptr;
}

void DefaultAllocator::Free(void* ptr) { free(ptr); }

void* SystemAllocator::Allocate(size_t size, size_t alignment) { return DefaultAllocator::Allocate(size, std::max<size_t>(alignment, 64)); }

void SystemAllocator::Free(void* ptr) { free(ptr); }

void LargeMemCopy(void* __restrict dest, const void* __restrict src, size_t size) { memcpy(dest, src, size); }   // (the reference's is a non-temporal copy)

} // namespace rt
