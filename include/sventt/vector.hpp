// include/sventt/vector.hpp -- caller-side buffers.
//
// PageMemory<T>(length, allocate_huge_pages) keeps the interface of the
// reference's class of the same name (include/sventt/vector.hpp:61-168 there):
// page-aligned anonymous memory, size()/data()/operator[]/at()/begin()/end(),
// reset().  On this build it is host memory that NTT::compute_* accepts (the engine
// stages host buffers through the device) as well as device pointers; mappings of up to
// 4 GiB are page-locked through the engine (sventt_host_register, include/sventt_hip.h) so
// that the staging copies run at the PCIe rate -- best effort, SVENTT_PIN_HOST=0 turns it
// off, =1 lifts the size limit.  AuxiliaryVector / FakeByteVector / pointer_utility keep the interface of
// the reference's byte blob (vector.hpp:24-48,170-254 there) for code that drives a
// kernel_type directly (prepare_forward(vec), compute_forward(dst, src, cursor)); on
// this build the blob holds plan records (plan_handle.hpp), the twiddle tables
// themselves live on the device and are owned by the plan.
#ifndef SVENTT_GPU_VECTOR_HPP_INCLUDED
#define SVENTT_GPU_VECTOR_HPP_INCLUDED

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <new>
#include <stdexcept>

#include <cstdlib>

#include <sys/mman.h>

// (weak: a program that only uses the containers need not link the engine)
extern "C" {
int sventt_host_register(void *host, std::size_t bytes) __attribute__((weak));
int sventt_host_unregister(void *host) __attribute__((weak));
}

namespace sventt {

template <class value_type_> class PageMemory {
public:
  using value_type = value_type_;
  using size_type = std::uint64_t;

private:
  size_type length{};
  size_type mapped_bytes{};
  value_type *base{};
  bool pinned{};

  void release(void) {
    if (base != nullptr) {
      if (pinned && sventt_host_unregister != nullptr) {
        (void)sventt_host_unregister(base);
      }
      munmap(base, mapped_bytes);
    }
    base = nullptr;
    mapped_bytes = 0;
    length = 0;
    pinned = false;
  }

  static bool want_pinning(const size_type bytes) {
    if (sventt_host_register == nullptr) {
      return false;
    }
    const char *const e{std::getenv("SVENTT_PIN_HOST")};
    if (e != nullptr) {
      return std::atoi(e) != 0;
    }
    return bytes <= (size_type{4} << 30);
  }

public:
  PageMemory(void) = default;

  PageMemory(const size_type len, const bool allocate_huge_pages) {
    reset(len, allocate_huge_pages);
  }

  PageMemory(const PageMemory &) = delete;
  PageMemory &operator=(const PageMemory &) = delete;

  PageMemory(PageMemory &&that) noexcept
      : length{that.length}, mapped_bytes{that.mapped_bytes}, base{that.base}, pinned{that.pinned} {
    that.base = nullptr;
    that.length = that.mapped_bytes = 0;
    that.pinned = false;
  }

  PageMemory &operator=(PageMemory &&that) noexcept {
    if (this != &that) {
      release();
      length = that.length;
      mapped_bytes = that.mapped_bytes;
      base = that.base;
      pinned = that.pinned;
      that.base = nullptr;
      that.length = that.mapped_bytes = 0;
      that.pinned = false;
    }
    return *this;
  }

  ~PageMemory(void) { release(); }

  size_type size(void) const { return length; }
  value_type *data(void) { return base; }
  const value_type *data(void) const { return base; }

  void reset(void) { release(); }

  // Huge pages are a request, not a requirement: when the kernel has none
  // reserved the mapping falls back to transparent huge pages.
  void reset(const size_type len, const bool allocate_huge_pages = false) {
    release();
    if (len == 0) {
      return;
    }
    const size_type granule{allocate_huge_pages ? (size_type{1} << 21) : (size_type{1} << 12)};
    const size_type bytes{(sizeof(value_type) * len + granule - 1) / granule * granule};
    void *p{MAP_FAILED};
    if (allocate_huge_pages) {
      p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE,
               MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_HUGETLB, -1, 0);
    }
    if (p == MAP_FAILED) {
      p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE,
               -1, 0);
      if (p != MAP_FAILED && allocate_huge_pages) {
        madvise(p, bytes, MADV_HUGEPAGE);
      }
    }
    if (p == MAP_FAILED) {
      throw std::bad_alloc{};
    }
    base = static_cast<value_type *>(p);
    mapped_bytes = bytes;
    length = len;
    pinned = want_pinning(bytes) && sventt_host_register(p, bytes) == 0;
  }

  value_type &operator[](const size_type index) { return base[index]; }
  const value_type &operator[](const size_type index) const { return base[index]; }

  value_type &at(const size_type index) {
    if (index >= length) {
      throw std::out_of_range{"Index out of range"};
    }
    return base[index];
  }

  const value_type &at(const size_type index) const {
    if (index >= length) {
      throw std::out_of_range{"Index out of range"};
    }
    return base[index];
  }

  value_type *begin(void) { return base; }
  value_type *end(void) { return base + length; }
  const value_type *begin(void) const { return base; }
  const value_type *end(void) const { return base + length; }
  const value_type *cbegin(void) const { return base; }
  const value_type *cend(void) const { return base + length; }
};

// ---- the auxiliary byte blob and its cursor helpers -----------------------------------
namespace pointer_utility {

// bytes to skip so that an object of type T can sit at offset/address `pos`
template <class T> std::size_t get_padding(const std::size_t pos) {
  const std::size_t misalignment{pos % alignof(T)};
  return misalignment == 0 ? 0 : alignof(T) - misalignment;
}

template <class T> std::size_t get_padding(const std::byte *const pointer) {
  return get_padding<T>(reinterpret_cast<std::uintptr_t>(pointer));
}

template <class T> void skip_padding(const std::byte *&pointer) { pointer += get_padding<T>(pointer); }

// the T at the (aligned) cursor; the cursor moves past it
template <class T> const T &get_and_advance(const std::byte *&pointer) {
  skip_padding<T>(pointer);
  const T *const object{reinterpret_cast<const T *>(pointer)};
  pointer += sizeof(T);
  return *object;
}

template <class T> const T &get(const std::byte *pointer) { return get_and_advance<T>(pointer); }

} // namespace pointer_utility

// Size-only dry run of prepare_*: counts what an AuxiliaryVector would hold.
class FakeByteVector {
public:
  using value_type = std::byte;
  using size_type = std::uint64_t;

private:
  size_type bytes{};

public:
  template <class T> void push_back([[maybe_unused]] const T &value) {
    bytes += pointer_utility::get_padding<T>(bytes) + sizeof(T);
  }

  size_type size(void) const { return bytes; }

  template <class T> T &reinterpret_at(const size_type index) {
    if (index + sizeof(T) > size()) {
      throw std::out_of_range{"Index out of range"};
    }
    static T scratch;
    return scratch;
  }
};

// Fixed-capacity byte vector that prepare_* appends to (std::bad_alloc when full).
class AuxiliaryVector {
public:
  using value_type = std::byte;
  using size_type = PageMemory<value_type>::size_type;

private:
  size_type used{};
  PageMemory<value_type> storage;

public:
  AuxiliaryVector(void) = default;

  AuxiliaryVector(const size_type capacity, const bool allocate_huge_pages = false)
      : storage{capacity, allocate_huge_pages} {}

  size_type size(void) const { return used; }
  size_type capacity(void) const { return storage.size(); }
  value_type *data(void) { return storage.data(); }
  const value_type *data(void) const { return storage.data(); }
  value_type &operator[](const size_type index) { return storage[index]; }
  const value_type &operator[](const size_type index) const { return storage[index]; }

  value_type &at(const size_type index) {
    if (index >= size()) {
      throw std::out_of_range{"Index out of range"};
    }
    return storage[index];
  }

  template <class T> T &reinterpret_at(const size_type index) {
    if (index + sizeof(T) > size()) {
      throw std::out_of_range{"Index out of range"};
    }
    return *reinterpret_cast<T *>(&storage[index]);
  }

  template <class T> void push_back(const T &value) {
    const size_type at{used + pointer_utility::get_padding<T>(used)};
    if (at + sizeof(T) > capacity()) {
      throw std::bad_alloc{};
    }
    std::memcpy(&storage[at], &value, sizeof(T));
    used = at + sizeof(T);
  }
};

} // namespace sventt

#endif /* SVENTT_GPU_VECTOR_HPP_INCLUDED */
