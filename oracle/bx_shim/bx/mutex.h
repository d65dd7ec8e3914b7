// Stand-in for <bx/mutex.h>: the reference guards its buffer pools with bx::Mutex (vg.cpp:383-385, 5007-5193);
// the oracle is single-threaded, so the lock is a no-op. Test infrastructure only.
#ifndef BX_SHIM_MUTEX_H
#define BX_SHIM_MUTEX_H
#include "bx.h"
namespace bx
{
class Mutex { public: void lock() {} void unlock() {} };
class MutexScope { public: explicit MutexScope(Mutex& m) : m_mutex(m) { m_mutex.lock(); } ~MutexScope() { m_mutex.unlock(); } private: Mutex& m_mutex; };
}
#endif
