// h5lite: a minimal reader / writer for the subset of HDF5 the reference's snapshots use
// (src/io/read_write_hdf5.rs:38-188: contiguous, un-chunked f64 (and 64-bit integer scalar) datasets of rank 1 or 2 inside
// one level of groups; src/navier_stokes/navier_io.rs:21-62, src/field/io.rs:74-110).
// There is no libhdf5 in this image, so the on-disk structures are produced and parsed by hand,
// following the HDF5 File Format Specification, "classic" layout -- the one libhdf5's default
// (libver earliest, what the hdf5 0.8.1 crate of the reference uses) writes:
//   superblock version 0, version-1 object headers, old-style groups (symbol-table message ->
//   version-1 B-tree node -> symbol-table node + local heap), dataspace v1, IEEE f64 LE datatype,
//   data layout v3 (contiguous).  No checksums exist in these versions.
// The writer always emits a complete file (existing datasets of a file are read, merged,
// rewritten: the reference's "create or append, overwrite when present" semantics).  The reader
// also accepts superblock version 1 and files whose groups spread over several B-tree / symbol
// nodes, i.e. what libhdf5 itself writes for these snapshots.  Host code only.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace rpde {
namespace h5 {

struct Dataset {
  std::vector<uint64_t> dims;   // rank 1 or 2
  std::vector<double> data;     // row-major
  // stored on disk as unsigned 64-bit little-endian integers (H5T_STD_U64LE: what the hdf5 crate writes for a Rust
  // `usize`, e.g. `num_save` of statistics.rs:153); the values are carried as doubles in memory
  bool u64 = false;
};

// datasets by path: "time", "ux/v", "temp/vhat_re" ... (at most one group level)
using Tree = std::map<std::string, Dataset>;

// write a complete classic-format HDF5 file holding `tree` (overwrites `filename`)
void write_file(const std::string& filename, const Tree& tree);
// create or append: datasets already in `filename` are kept unless `tree` names them again
void update_file(const std::string& filename, const Tree& tree);

class Reader {
 public:
  explicit Reader(const std::string& filename);
  ~Reader();
  Reader(const Reader&) = delete;
  Reader& operator=(const Reader&) = delete;
  bool has(const std::string& path) const { return index_.count(path) != 0; }
  std::vector<std::string> paths() const;
  std::vector<uint64_t> shape(const std::string& path) const;
  Dataset read(const std::string& path) const;

 private:
  struct Entry { std::vector<uint64_t> dims; uint64_t addr = 0, bytes = 0; bool compact = false; std::vector<uint8_t> inline_data;
                 int kind = 0; };   // kind: 0 = f64, 1 = u64, 2 = i64
  void walk_group(uint64_t oh_addr, const std::string& prefix, int depth);
  void walk_btree(uint64_t node, uint64_t heap_data, const std::string& prefix, int depth);
  void parse_object(uint64_t oh_addr, const std::string& path, int depth);
  void pread_(void* dst, uint64_t off, uint64_t n) const;
  struct Msg { uint16_t type; std::vector<uint8_t> data; };
  std::vector<Msg> object_messages(uint64_t oh_addr) const;
  std::string heap_string(uint64_t heap_data, uint64_t off) const;
  void* f_ = nullptr;
  uint64_t base_ = 0, size_ = 0;
  int leaf_k_ = 4, internal_k_ = 16;
  std::map<std::string, Entry> index_;
};

}  // namespace h5
}  // namespace rpde
