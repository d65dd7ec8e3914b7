// Stand-in for <bx/handlealloc.h>: dense/sparse handle allocator with bx's semantics (alloc() hands out the
// lowest never-used slot first and recycles freed ones LIFO; kInvalidHandle when full). vg.cpp uses it for image
// and command-list handles (vg.cpp:781-782, 2308, 2360, 5470, 5657, 5671). Test infrastructure only.
#ifndef BX_SHIM_HANDLEALLOC_H
#define BX_SHIM_HANDLEALLOC_H
#include "allocator.h"
namespace bx
{
constexpr uint16_t kInvalidHandle = UINT16_MAX;
class HandleAlloc
{
public:
	explicit HandleAlloc(uint16_t maxHandles) : m_numHandles(0), m_maxHandles(maxHandles)
	{
		uint16_t* dense = getDensePtr();
		for (uint16_t i = 0; i < m_maxHandles; ++i) { dense[i] = i; }
	}
	const uint16_t* getHandles() const { return getDensePtr(); }
	uint16_t getHandleAt(uint16_t at) const { return getDensePtr()[at]; }
	uint16_t getNumHandles() const { return m_numHandles; }
	uint16_t getMaxHandles() const { return m_maxHandles; }
	uint16_t alloc()
	{
		if (m_numHandles < m_maxHandles) {
			const uint16_t index = m_numHandles++;
			uint16_t* dense = getDensePtr();
			const uint16_t handle = dense[index];
			getSparsePtr()[handle] = index;
			return handle;
		}
		return kInvalidHandle;
	}
	bool isValid(uint16_t handle) const
	{
		const uint16_t index = getSparsePtr()[handle];
		return index < m_numHandles && getDensePtr()[index] == handle;
	}
	void free(uint16_t handle)
	{
		uint16_t* dense = getDensePtr();
		uint16_t* sparse = getSparsePtr();
		const uint16_t index = sparse[handle];
		--m_numHandles;
		const uint16_t temp = dense[m_numHandles];
		dense[m_numHandles] = handle;
		sparse[temp] = index;
		dense[index] = temp;
	}
	void reset() { m_numHandles = 0; uint16_t* dense = getDensePtr(); for (uint16_t i = 0; i < m_maxHandles; ++i) { dense[i] = i; } }
private:
	uint16_t* getDensePtr() const { return (uint16_t*)((uint8_t*)this + sizeof(HandleAlloc)); }
	uint16_t* getSparsePtr() const { return getDensePtr() + m_maxHandles; }
	uint16_t m_numHandles;
	uint16_t m_maxHandles;
};
inline HandleAlloc* createHandleAlloc(AllocatorI* a, uint16_t maxHandles)
{
	uint8_t* p = (uint8_t*)bx::alloc(a, sizeof(HandleAlloc) + 2 * maxHandles * sizeof(uint16_t));
	return ::new (p) HandleAlloc(maxHandles);
}
inline void destroyHandleAlloc(AllocatorI* a, HandleAlloc* h) { h->~HandleAlloc(); bx::free(a, h); }
}
#endif
