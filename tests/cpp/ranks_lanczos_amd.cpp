// A user program on the header-only solver API, run as ONE PROCESS PER RANK without MPI or torch:
//   ranks_lanczos_amd <rank> <world> <device> <id_file> <grid_edge> <iterations>
// Rank 0 asks the library for a communicator id (eigenex_rccl_unique_id) and publishes it in <id_file>; the others wait for
// the file.  Every rank then builds its rows of the 7-point Laplacian on a grid_edge^3 grid in CSR (global column indices),
// hands them to device::CsrOperator and calls LanczosEigenSolver<double>::compute() exactly as a single-GPU program would.
// Prints one JSON line: eigenvalues (the same on every rank) and this rank's rows of the first Ritz vector.
// world = 1 runs the same program on a plain single-GPU context (the comparison the test makes).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "cmpt/eigen_ex/lanczos.hpp"

int main(int argc, char** argv) {
  using namespace cmpt::EigenEx;
  if (argc != 7) return 2;
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]), dev = std::atoi(argv[3]);
  const char* id_file = argv[4];
  const std::int64_t n = std::atoll(argv[5]), N = n * n * n;
  const int iterations = std::atoi(argv[6]);
  try {
    std::shared_ptr<device::Context> ctx;
    if (world == 1) {
      ctx = std::make_shared<device::Context>(dev);
    } else {
      unsigned char id[128];
      if (rank == 0) {
        device::check(eigenex_rccl_unique_id(id), "eigenex_rccl_unique_id");
        const std::string tmp = std::string(id_file) + ".part";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) return 3;
        std::fclose(f);
        std::rename(tmp.c_str(), id_file);
      } else {
        FILE* f = nullptr;
        for (int tries = 0; !(f = std::fopen(id_file, "rb")); ++tries) {
          if (tries > 60000) return 4;
          std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        if (std::fread(id, 1, sizeof id, f) != sizeof id) return 5;
        std::fclose(f);
      }
      ctx = std::make_shared<device::Context>(dev, rank, world, id);
    }
    // this rank's rows: the library's own partition rule
    std::int64_t rb = 0, re = N;
    if (world > 1) device::check(eigenex_partition(N, world, rank, &rb, &re), "eigenex_partition");
    std::vector<std::int32_t> rowptr(1, 0), col;
    std::vector<double> val;
    for (std::int64_t i = rb; i < re; ++i) {
      const std::int64_t x = i % n, y = (i / n) % n, z = i / (n * n);
      if (z > 0) col.push_back((std::int32_t)(i - n * n)), val.push_back(-1.0);
      if (y > 0) col.push_back((std::int32_t)(i - n)), val.push_back(-1.0);
      if (x > 0) col.push_back((std::int32_t)(i - 1)), val.push_back(-1.0);
      col.push_back((std::int32_t)i), val.push_back(6.0);
      if (x < n - 1) col.push_back((std::int32_t)(i + 1)), val.push_back(-1.0);
      if (y < n - 1) col.push_back((std::int32_t)(i + n)), val.push_back(-1.0);
      if (z < n - 1) col.push_back((std::int32_t)(i + n * n)), val.push_back(-1.0);
      rowptr.push_back((std::int32_t)col.size());
    }
    auto op = std::make_shared<device::CsrOperator>(ctx, N, rb, re - rb, rowptr.data(), col.data(), val.data());
    LanczosEigenSolver<double> es;
    es.setDeviceOperator(op).setMinIterations(iterations).setMaxIterations(iterations).setMaxEigenvalues(3);
    es.compute();
    std::printf("{\"rank\": %d, \"rows\": [%lld, %lld], \"info\": %d, \"iterations\": %ld, \"eigenvalues\": [", rank, (long long)rb, (long long)re,
                (int)es.info(), (long)es.iterations());
    for (Index i = 0; i < es.eigenvalues().size(); ++i) std::printf("%s%.17g", i ? ", " : "", es.eigenvalues()[i]);
    std::printf("], \"vector_rows\": %ld, \"first_vector\": [", (long)es.eigenvectors().rows());
    for (Index r = 0; r < es.eigenvectors().rows(); ++r) std::printf("%s%.17g", r ? ", " : "", es.eigenvectors()(r, 0));
    std::printf("]}\n");
  } catch (const std::exception& e) {
    std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
    return 1;
  }
  return 0;
}
