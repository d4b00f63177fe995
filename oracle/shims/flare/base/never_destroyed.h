// Oracle shim (test infrastructure, NOT product code). task_dispatcher.cc:74-75.
#ifndef ORACLE_SHIM_FLARE_NEVER_DESTROYED_H_
#define ORACLE_SHIM_FLARE_NEVER_DESTROYED_H_
#include <new>
#include <utility>
namespace flare {
template <class T>
class NeverDestroyed {
 public:
  template <class... Args>
  NeverDestroyed(Args&&... args) {
    new (storage_) T(std::forward<Args>(args)...);
  }
  T* Get() { return reinterpret_cast<T*>(storage_); }
  T* operator->() { return Get(); }

 private:
  alignas(T) unsigned char storage_[sizeof(T)];
};
}  // namespace flare
#endif
