// RAII views of the C ABI (include/eigenex_hip.h) for the header-only solver
// classes.  Every failure of the HIP library becomes a LanczosException: there
// is no host fallback for the vector work.
#pragma once

#include <complex>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "eigenex_hip.h"

namespace cmpt {
namespace EigenEx {

// exception class of the reference (lanczos.hpp:90-95); ArnoldiException is an alias (arnoldi.hpp:45)
class LanczosException : public std::runtime_error {
 public:
  explicit LanczosException(const char* message) : std::runtime_error(message) {}
  explicit LanczosException(const std::string& message) : std::runtime_error(message) {}
};

namespace device {

inline void check(int rc, const char* what) {
  if (rc != 0) throw LanczosException(std::string(what) + ": " + eigenex_last_error());
}

// One GPU / one shard per process (RCCL between processes), or the in-process
// loopback partition used for verification.
class Context {
 public:
  // single GPU
  explicit Context(int device_index = 0) { check(eigenex_context_create(device_index, 0, 1, nullptr, &h_), "eigenex_context_create"); }
  // one process per GPU: rccl_id from eigenex_rccl_unique_id() on rank 0, broadcast by the launcher
  Context(int device_index, int rank, int world_size, const void* rccl_id128) {
    check(eigenex_context_create(device_index, rank, world_size, rccl_id128, &h_), "eigenex_context_create");
  }
  struct Loopback {
    int shards;
  };
  Context(int device_index, Loopback lb) { check(eigenex_context_create_loopback(device_index, lb.shards, &h_), "eigenex_context_create_loopback"); }
  // adopt a handle created elsewhere (e.g. by a ctypes/cgo host); not owned
  static std::shared_ptr<Context> borrow(eigenex_context_t h) {
    std::shared_ptr<Context> c(new Context(h));
    return c;
  }
  ~Context() {
    if (owned_ && h_) eigenex_context_destroy(h_);
  }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  eigenex_context_t handle() const { return h_; }
  void sync() const { check(eigenex_context_sync(h_), "eigenex_context_sync"); }
  int rank() const { return info(0); }
  int worldSize() const { return info(1); }
  int shardsTotal() const { return info(2); }
  int shardsLocal() const { return info(3); }

 private:
  explicit Context(eigenex_context_t h) : h_(h), owned_(false) {}
  int info(int which) const {
    int v[4];
    check(eigenex_context_info(h_, &v[0], &v[1], &v[2], &v[3]), "eigenex_context_info");
    return v[which];
  }
  eigenex_context_t h_ = nullptr;
  bool owned_ = true;
};

// Device-resident CSR operator: what stands behind setMatrixMultiplication
// (lanczos.hpp:179-188) when the operator is a sparse matrix.
class CsrOperator {
 public:
  // rows owned by this context, global column indices (see eigenex_csr_upload)
  CsrOperator(std::shared_ptr<Context> ctx, std::int64_t n_global, std::int64_t row_begin, std::int64_t n_rows,
              const std::int32_t* rowptr, const std::int32_t* col_global, const double* val)
      : ctx_(std::move(ctx)), n_(n_global) {
    check(eigenex_csr_upload(ctx_->handle(), n_global, row_begin, n_rows, rowptr, col_global, val, &h_), "eigenex_csr_upload");
  }
  // 64-bit row pointers (the reference's Index, lanczos.hpp:108-116): a shard may hold >= 2^31 stored entries (eigenex_csr_upload64)
  CsrOperator(std::shared_ptr<Context> ctx, std::int64_t n_global, std::int64_t row_begin, std::int64_t n_rows,
              const std::int64_t* rowptr, const std::int32_t* col_global, const double* val)
      : ctx_(std::move(ctx)), n_(n_global) {
    check(eigenex_csr_upload64(ctx_->handle(), n_global, row_begin, n_rows, rowptr, col_global, val, &h_), "eigenex_csr_upload64");
  }
  // fp32 values (a Scalar = float user's matrix): widened here, the device stores and multiplies in fp64
  CsrOperator(std::shared_ptr<Context> ctx, std::int64_t n_global, std::int64_t row_begin, std::int64_t n_rows,
              const std::int32_t* rowptr, const std::int32_t* col_global, const float* val)
      : ctx_(std::move(ctx)), n_(n_global) {
    const std::vector<double> wide(val, val + (n_rows > 0 ? rowptr[n_rows] - rowptr[0] : 0));
    check(eigenex_csr_upload(ctx_->handle(), n_global, row_begin, n_rows, rowptr, col_global, wide.data(), &h_), "eigenex_csr_upload");
  }
  // complex values (std::complex<double>, crossing the C ABI as interleaved doubles)
  static std::shared_ptr<CsrOperator> complexCsr(std::shared_ptr<Context> ctx, std::int64_t n_global, std::int64_t row_begin,
                                                 std::int64_t n_rows, const std::int32_t* rowptr,
                                                 const std::int32_t* col_global, const std::complex<double>* val) {
    std::shared_ptr<CsrOperator> op(new CsrOperator());
    op->ctx_ = std::move(ctx);
    op->n_ = n_global;
    check(eigenex_csr_upload_z(op->ctx_->handle(), n_global, row_begin, n_rows, rowptr, col_global,
                               reinterpret_cast<const double*>(val), &op->h_),
          "eigenex_csr_upload_z");
    return op;
  }
  // synthetic 7-point Laplacian on an n^3 grid, generated on the device
  static std::shared_ptr<CsrOperator> laplacian3d(std::shared_ptr<Context> ctx, std::int64_t n) {
    std::shared_ptr<CsrOperator> op(new CsrOperator());
    op->ctx_ = std::move(ctx);
    op->n_ = n * n * n;
    check(eigenex_csr_laplacian3d(op->ctx_->handle(), n, &op->h_), "eigenex_csr_laplacian3d");
    return op;
  }
  static std::shared_ptr<CsrOperator> borrow(std::shared_ptr<Context> ctx, eigenex_csr_t h) {
    std::shared_ptr<CsrOperator> op(new CsrOperator());
    op->ctx_ = std::move(ctx);
    op->h_ = h;
    op->owned_ = false;
    check(eigenex_csr_info(h, &op->n_, nullptr, nullptr, nullptr), "eigenex_csr_info");
    return op;
  }
  // takes ownership of a handle made by one of the C entry points (eigenex_block_upload, eigenex_csr_upload_ex, ...)
  static std::shared_ptr<CsrOperator> adopt(std::shared_ptr<Context> ctx, eigenex_csr_t h) {
    std::shared_ptr<CsrOperator> op = borrow(std::move(ctx), h);
    op->owned_ = true;
    return op;
  }
  ~CsrOperator() {
    if (owned_ && h_) eigenex_csr_destroy(h_);
  }
  CsrOperator(const CsrOperator&) = delete;
  CsrOperator& operator=(const CsrOperator&) = delete;
  eigenex_csr_t handle() const { return h_; }
  const std::shared_ptr<Context>& context() const { return ctx_; }
  std::int64_t rows() const { return n_; }
  std::int64_t localRows() const {
    std::int64_t nl = 0;
    check(eigenex_csr_info(h_, nullptr, &nl, nullptr, nullptr), "eigenex_csr_info");
    return nl;
  }

 private:
  CsrOperator() = default;
  std::shared_ptr<Context> ctx_;
  eigenex_csr_t h_ = nullptr;
  std::int64_t n_ = 0;
  bool owned_ = true;
};

// process-wide default context for solvers that are only given a host callback
inline std::shared_ptr<Context> defaultContext() {
  static std::shared_ptr<Context> ctx = std::make_shared<Context>(0);
  return ctx;
}

}  // namespace device
}  // namespace EigenEx
}  // namespace cmpt
