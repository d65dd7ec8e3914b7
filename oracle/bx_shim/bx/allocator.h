// Stand-in for <bx/allocator.h>: AllocatorI + the free-function helpers the reference calls
// (path.cpp:25,35-38,50,67,752; stroker.cpp:196,209-229,2325-2339). Test infrastructure only.
#ifndef BX_SHIM_ALLOCATOR_H
#define BX_SHIM_ALLOCATOR_H

#include "bx.h"
#include <stdlib.h>
#include <new>

namespace bx
{
struct AllocatorI
{
	virtual ~AllocatorI() {}
	virtual void* realloc(void* ptr, size_t size, size_t align, const char* file, uint32_t line) = 0;
};

// Plain malloc-backed allocator with a header that remembers the unaligned base.
struct ShimAllocator : public AllocatorI
{
	void* realloc(void* ptr, size_t size, size_t align, const char*, uint32_t) override
	{
		if (align < sizeof(void*) * 2) { align = sizeof(void*) * 2; }
		if (size == 0) {
			if (ptr) { ::free(((void**)ptr)[-2]); }
			return nullptr;
		}
		const size_t total = size + align + sizeof(void*) * 2;
		uint8_t* base = (uint8_t*)::calloc(1, total); // zeroed: padding bytes of command lists / pooled vertex buffers are deterministic
		uintptr_t p = ((uintptr_t)base + sizeof(void*) * 2 + (align - 1)) & ~(uintptr_t)(align - 1);
		void** hdr = (void**)p;
		hdr[-2] = base;
		hdr[-1] = (void*)size;
		if (ptr) {
			const size_t old = (size_t)((void**)ptr)[-1];
			::memcpy((void*)p, ptr, old < size ? old : size);
			::free(((void**)ptr)[-2]);
		}
		return (void*)p;
	}
};

inline void* alloc(AllocatorI* a, size_t size, size_t align = 0) { return a->realloc(nullptr, size, align, "", 0); }
inline void free(AllocatorI* a, void* ptr, size_t align = 0) { if (ptr) { a->realloc(ptr, 0, align, "", 0); } }
inline void* realloc(AllocatorI* a, void* ptr, size_t size, size_t align = 0) { return a->realloc(ptr, size, align, "", 0); }
inline void* alignedAlloc(AllocatorI* a, size_t size, size_t align) { return a->realloc(nullptr, size, align, "", 0); }
inline void alignedFree(AllocatorI* a, void* ptr, size_t align) { if (ptr) { a->realloc(ptr, 0, align, "", 0); } }
inline void* alignedRealloc(AllocatorI* a, void* ptr, size_t size, size_t align) { return a->realloc(ptr, size, align, "", 0); }
template<typename T> inline void deleteObject(AllocatorI* a, T* obj, size_t align = 0) { if (obj) { obj->~T(); bx::free(a, obj, align); } }
}
#define BX_NEW(_allocator, _type) ::new (bx::alloc(_allocator, sizeof(_type))) _type

#endif
