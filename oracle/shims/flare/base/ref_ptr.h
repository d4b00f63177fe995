// Oracle shim (test infrastructure, NOT product code).
// Intrusive ref counting as used at task_dispatcher.h:184,196,202 and
// .cc:132,206,235. Single-threaded harness, so a plain counter is enough.
#ifndef ORACLE_SHIM_FLARE_REF_PTR_H_
#define ORACLE_SHIM_FLARE_REF_PTR_H_
#include <cstddef>
#include <utility>
namespace flare {
template <class T>
class RefCounted {
 public:
  void Ref() const { ++refs_; }
  void Deref() const {
    if (--refs_ == 0) delete static_cast<const T*>(this);
  }

 protected:
  RefCounted() = default;
  ~RefCounted() = default;

 private:
  mutable std::size_t refs_ = 1;  // Born owned by its creator.
};

struct ref_ptr_t {};
struct adopt_ptr_t {};
inline constexpr ref_ptr_t ref_ptr{};
inline constexpr adopt_ptr_t adopt_ptr{};

template <class T>
class RefPtr {
 public:
  RefPtr() = default;
  RefPtr(std::nullptr_t) {}
  RefPtr(ref_ptr_t, T* p) : p_(p) {
    if (p_) p_->Ref();
  }
  RefPtr(adopt_ptr_t, T* p) : p_(p) {}
  RefPtr(const RefPtr& o) : p_(o.p_) {
    if (p_) p_->Ref();
  }
  RefPtr(RefPtr&& o) noexcept : p_(o.p_) { o.p_ = nullptr; }
  RefPtr& operator=(RefPtr o) noexcept {
    std::swap(p_, o.p_);
    return *this;
  }
  ~RefPtr() {
    if (p_) p_->Deref();
  }
  T* Get() const { return p_; }
  T* operator->() const { return p_; }
  T& operator*() const { return *p_; }
  explicit operator bool() const { return p_ != nullptr; }

 private:
  T* p_ = nullptr;
};

template <class T, class... Args>
RefPtr<T> MakeRefCounted(Args&&... args) {
  return RefPtr<T>(adopt_ptr, new T(std::forward<Args>(args)...));
}
}  // namespace flare
#endif
