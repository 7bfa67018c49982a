// include/sventt/vector.hpp -- caller-side buffers.
//
// PageMemory<T>(length, allocate_huge_pages) keeps the interface of the
// reference's class of the same name (include/sventt/vector.hpp:61-168 there):
// page-aligned anonymous memory, size()/data()/operator[]/at()/begin()/end(),
// reset().  On this build it is ordinary host memory; NTT::compute_* accepts it
// (the engine stages host buffers through the device) as well as device
// pointers.  The reference's AuxiliaryVector / FakeByteVector carried its
// SVE-width-dependent twiddle blob and have no counterpart: the device tables are
// owned by the plan ("not part of the contract").
#ifndef SVENTT_GPU_VECTOR_HPP_INCLUDED
#define SVENTT_GPU_VECTOR_HPP_INCLUDED

#include <cstddef>
#include <cstdint>
#include <new>
#include <stdexcept>

#include <sys/mman.h>

namespace sventt {

template <class value_type_> class PageMemory {
public:
  using value_type = value_type_;
  using size_type = std::uint64_t;

private:
  size_type length{};
  size_type mapped_bytes{};
  value_type *base{};

  void release(void) {
    if (base != nullptr) {
      munmap(base, mapped_bytes);
    }
    base = nullptr;
    mapped_bytes = 0;
    length = 0;
  }

public:
  PageMemory(void) = default;

  PageMemory(const size_type len, const bool allocate_huge_pages) {
    reset(len, allocate_huge_pages);
  }

  PageMemory(const PageMemory &) = delete;
  PageMemory &operator=(const PageMemory &) = delete;

  PageMemory(PageMemory &&that) noexcept
      : length{that.length}, mapped_bytes{that.mapped_bytes}, base{that.base} {
    that.base = nullptr;
    that.length = that.mapped_bytes = 0;
  }

  PageMemory &operator=(PageMemory &&that) noexcept {
    if (this != &that) {
      release();
      length = that.length;
      mapped_bytes = that.mapped_bytes;
      base = that.base;
      that.base = nullptr;
      that.length = that.mapped_bytes = 0;
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

} // namespace sventt

#endif /* SVENTT_GPU_VECTOR_HPP_INCLUDED */
