#include "h5lite.h"

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>

#include "platform.h"

namespace rpde {
namespace h5 {
namespace {

constexpr uint64_t kUndef = ~0ULL;
constexpr uint8_t kSig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
constexpr int kInternalK = 16;   // group B-tree internal node K (superblock field; libhdf5 default)

// ---------------------------------------------------------------------------------- writer
struct Buf {
  std::vector<uint8_t> b;
  uint64_t base = 0;                 // file address of b[0]: size() and patch64() speak file addresses
  uint64_t size() const { return base + b.size(); }
  void align8() { while (size() % 8) b.push_back(0); }
  void u8(uint8_t v) { b.push_back(v); }
  void u16(uint16_t v) { for (int i = 0; i < 2; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
  void u32(uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
  void u64(uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
  void zeros(size_t n) { b.insert(b.end(), n, 0); }
  void bytes(const void* p, size_t n) { const uint8_t* q = static_cast<const uint8_t*>(p); b.insert(b.end(), q, q + n); }
  void patch64(uint64_t at, uint64_t v) { for (int i = 0; i < 8; ++i) b[at - base + i] = (uint8_t)(v >> (8 * i)); }
};

// one version-1 header message: type, size of (8-byte padded) data, flags, 3 reserved bytes, data
void message(Buf& o, uint16_t type, const std::vector<uint8_t>& data, uint8_t flags = 0) {
  const size_t padded = (data.size() + 7) & ~size_t(7);
  o.u16(type); o.u16((uint16_t)padded); o.u8(flags); o.zeros(3);
  o.bytes(data.data(), data.size());
  o.zeros(padded - data.size());
}

// version-1 object header around `msgs` (already encoded by message()); returns its address
uint64_t object_header(Buf& f, const Buf& msgs, int nmsg) {
  f.align8();
  const uint64_t at = f.size();
  f.u8(1); f.u8(0); f.u16((uint16_t)nmsg); f.u32(1); f.u32((uint32_t)msgs.size());
  f.zeros(4);   // the 12-byte prefix is padded so that the messages start 8-byte aligned
  f.bytes(msgs.b.data(), msgs.b.size());
  return at;
}

uint64_t dataset_header(Buf& f, const Dataset& d, uint64_t data_addr) {
  Buf m;
  {  // dataspace, version 1: rank, no maximum dimensions
    Buf s; s.u8(1); s.u8((uint8_t)d.dims.size()); s.u8(0); s.u8(0); s.u32(0);
    for (uint64_t v : d.dims) s.u64(v);
    message(m, 0x0001, s.b);
  }
  if (d.u64) {  // datatype: class 0 (fixed point) version 1, unsigned, little endian, 8 bytes: bit offset 0, precision 64
    const uint8_t t[12] = {0x10, 0x00, 0x00, 0x00, 8, 0, 0, 0, 0, 0, 64, 0};
    message(m, 0x0003, std::vector<uint8_t>(t, t + 12), 0x01);
  } else {  // datatype: class 1 (floating point) version 1, IEEE binary64 little endian
    const uint8_t t[20] = {0x11, 0x20, 0x3f, 0x00, 8, 0, 0, 0,    // class|version, bit fields (LE, msb implied, sign at 63), size 8
                           0, 0, 64, 0, 52, 11, 0, 52, 0xff, 0x03, 0, 0};   // offset 0, precision 64, exp at 52 (11 bits), mantissa at 0 (52 bits), bias 1023
    message(m, 0x0003, std::vector<uint8_t>(t, t + 20), 0x01);
  }
  {  // fill value, version 2: allocation late, write time "if set", no fill value defined
    const uint8_t v[4] = {2, 2, 2, 0};
    message(m, 0x0005, std::vector<uint8_t>(v, v + 4));
  }
  {  // data layout, version 3, class 1 (contiguous): address, size
    Buf l; l.u8(3); l.u8(1); l.u64(data_addr); l.u64(d.data.size() * 8);
    message(m, 0x0008, l.b);
  }
  return object_header(f, m, 4);
}

struct Child { std::string name; uint64_t oh; };

// old-style group: object header (symbol table message) + B-tree leaf + symbol node + local heap
uint64_t group_header(Buf& f, std::vector<Child> kids, int leaf_k, uint64_t* btree_out, uint64_t* heap_out) {
  std::sort(kids.begin(), kids.end(), [](const Child& a, const Child& b) { return a.name < b.name; });
  RPDE_REQUIRE((int)kids.size() <= 2 * leaf_k, "h5lite: group has more entries than one symbol node holds");
  // local heap data segment: the empty string at offset 0, then the names, each padded to 8 bytes
  Buf seg;
  seg.zeros(8);
  std::vector<uint64_t> off;
  for (const Child& c : kids) {
    off.push_back(seg.size());
    seg.bytes(c.name.c_str(), c.name.size() + 1);
    seg.align8();
  }
  const uint64_t free_off = seg.size();
  seg.u64(1); seg.u64(16);          // one free block closes the segment: next = 1 (end of list), 16 bytes long
  f.align8();
  const uint64_t heap_data = f.size();
  f.bytes(seg.b.data(), seg.b.size());
  const uint64_t heap = f.size();
  f.bytes("HEAP", 4); f.u8(0); f.zeros(3);
  f.u64(seg.size()); f.u64(free_off); f.u64(heap_data);
  // symbol table node
  const uint64_t snod = f.size();
  f.bytes("SNOD", 4); f.u8(1); f.u8(0); f.u16((uint16_t)kids.size());
  for (size_t i = 0; i < (size_t)(2 * leaf_k); ++i) {
    if (i < kids.size()) { f.u64(off[i]); f.u64(kids[i].oh); f.u32(0); f.u32(0); f.zeros(16); }
    else f.zeros(40);
  }
  // B-tree leaf node (type 0 = group): key 0 = "", child 0 = the symbol node, key 1 = its largest name
  const uint64_t btree = f.size();
  f.bytes("TREE", 4); f.u8(0); f.u8(0); f.u16(kids.empty() ? 0 : 1);
  f.u64(kUndef); f.u64(kUndef);
  const size_t body = (size_t)(2 * kInternalK + 1) * 8 + (size_t)(2 * kInternalK) * 8;
  const uint64_t body_at = f.size();
  f.zeros(body);
  if (!kids.empty()) { f.patch64(body_at + 8, snod); f.patch64(body_at + 16, off.back()); }
  // object header with the symbol table message
  Buf m;
  { Buf s; s.u64(btree); s.u64(heap); message(m, 0x0011, s.b); }
  *btree_out = btree; *heap_out = heap;
  return object_header(f, m, 1);
}

}  // namespace

void write_file(const std::string& filename, const Tree& tree) {
  // group the datasets: "" = root
  std::map<std::string, std::vector<std::pair<std::string, const Dataset*>>> groups;
  groups[""];
  for (const auto& kv : tree) {
    const std::string& path = kv.first;
    const size_t s = path.find('/');
    RPDE_REQUIRE(!path.empty() && path.find('/', s == std::string::npos ? 0 : s + 1) == std::string::npos,
                 "h5lite: paths have at most one group level: " + path);
    RPDE_REQUIRE(kv.second.dims.size() == 1 || kv.second.dims.size() == 2, "h5lite: rank 1 or 2 datasets only");
    uint64_t n = 1;
    for (uint64_t d : kv.second.dims) n *= d;
    RPDE_REQUIRE(n == kv.second.data.size(), "h5lite: shape and data length differ for " + path);
    if (s == std::string::npos) groups[""].push_back({path, &kv.second});
    else groups[path.substr(0, s)].push_back({path.substr(s + 1), &kv.second});
  }
  size_t most = groups[""].size() + groups.size() - 1;
  for (const auto& g : groups) most = std::max(most, g.second.size());
  const int leaf_k = std::max<int>(4, (int)(most + 1) / 2);

  Buf f;
  // superblock, version 0 (96 bytes); root entry and end-of-file address are patched at the end
  f.bytes(kSig, 8);
  f.u8(0); f.u8(0); f.u8(0); f.u8(0); f.u8(0); f.u8(8); f.u8(8); f.u8(0);
  f.u16((uint16_t)leaf_k); f.u16(kInternalK); f.u32(0);
  f.u64(0); f.u64(kUndef);
  const uint64_t eof_at = f.size(); f.u64(0);
  f.u64(kUndef);
  const uint64_t root_entry = f.size();
  f.u64(0); f.u64(0); f.u32(1); f.u32(0); f.u64(0); f.u64(0);
  RPDE_REQUIRE(f.size() == 96, "h5lite: superblock layout");

  // crash safety: everything goes to <filename>.tmp, which replaces the target by rename() only after a complete,
  // flushed write -- an interrupted write never costs the previous snapshot / statistics file
  // (a name of its own per writer: two writers of one target -- two processes, a callback and a statistics write -- must not
  // truncate each other's temporary file)
  static std::atomic<unsigned long> serial{0};
  const std::string tmpname = filename + ".tmp." + std::to_string((long)getpid()) + "." + std::to_string(serial.fetch_add(1));
  struct FileGuard {
    FILE* fp; std::string tmp;
    ~FileGuard() { if (fp) { std::fclose(fp); std::remove(tmp.c_str()); } }   // reached with fp set only on a throw
  } guard{std::fopen(tmpname.c_str(), "wb"), tmpname};
  FILE* fp = guard.fp;
  RPDE_REQUIRE(fp != nullptr, "h5lite: cannot create " + tmpname);
  // raw data first (addresses are needed by the dataset headers); streamed, not buffered
  std::map<const Dataset*, uint64_t> addr;
  uint64_t pos = 96;
  std::fseek(fp, 96, SEEK_SET);
  for (const auto& kv : tree) {
    addr[&kv.second] = pos;
    const size_t nb = kv.second.data.size() * 8;
    if (nb && kv.second.u64) {
      std::vector<uint64_t> iv(kv.second.data.size());
      for (size_t i = 0; i < iv.size(); ++i) iv[i] = (uint64_t)kv.second.data[i];
      RPDE_REQUIRE(std::fwrite(iv.data(), 1, nb, fp) == nb, "h5lite: short write");
    } else if (nb) RPDE_REQUIRE(std::fwrite(kv.second.data.data(), 1, nb, fp) == nb, "h5lite: short write");
    pos += nb;
  }
  // metadata after the data, built at its absolute file addresses
  RPDE_REQUIRE(pos % 8 == 0, "h5lite: data region must end 8-byte aligned");
  Buf g;
  g.base = pos;
  std::map<std::string, std::vector<Child>> kids_of;
  for (const auto& gkv : groups)
    for (const auto& d : gkv.second) kids_of[gkv.first].push_back(Child{d.first, dataset_header(g, *d.second, addr[d.second])});
  for (const auto& gkv : groups) {
    if (gkv.first.empty()) continue;
    uint64_t bt, hp;
    kids_of[""].push_back(Child{gkv.first, group_header(g, kids_of[gkv.first], leaf_k, &bt, &hp)});
  }
  uint64_t rbt = 0, rhp = 0;
  const uint64_t root_oh = group_header(g, kids_of[""], leaf_k, &rbt, &rhp);
  g.align8();
  const uint64_t eof = g.size();
  RPDE_REQUIRE(std::fwrite(g.b.data(), 1, g.b.size(), fp) == g.b.size(), "h5lite: short write");
  // superblock
  f.patch64(eof_at, eof);
  f.patch64(root_entry + 8, root_oh);
  f.patch64(root_entry + 24, rbt);
  f.patch64(root_entry + 32, rhp);
  std::fseek(fp, 0, SEEK_SET);
  RPDE_REQUIRE(std::fwrite(f.b.data(), 1, 96, fp) == 96, "h5lite: short write");
  RPDE_REQUIRE(std::fflush(fp) == 0, "h5lite: flush failed for " + tmpname);
  RPDE_REQUIRE(fsync(fileno(fp)) == 0, "h5lite: fsync failed for " + tmpname);
  guard.fp = nullptr;
  RPDE_REQUIRE(std::fclose(fp) == 0, "h5lite: close failed for " + tmpname);
  if (std::rename(tmpname.c_str(), filename.c_str()) != 0) {
    std::remove(tmpname.c_str());
    RPDE_REQUIRE(false, "h5lite: cannot move " + tmpname + " over " + filename);
  }
}

void update_file(const std::string& filename, const Tree& tree) {
  Tree all;
  if (FILE* t = std::fopen(filename.c_str(), "rb")) {
    std::fclose(t);
    Reader r(filename);
    for (const std::string& p : r.paths())
      if (!tree.count(p)) all[p] = r.read(p);
  }
  for (const auto& kv : tree) all[kv.first] = kv.second;
  write_file(filename, all);
}

// ---------------------------------------------------------------------------------- reader
static uint64_t le(const uint8_t* p, int n) {
  uint64_t v = 0;
  for (int i = 0; i < n; ++i) v |= (uint64_t)p[i] << (8 * i);
  return v;
}

Reader::Reader(const std::string& filename) {
  FILE* fp = std::fopen(filename.c_str(), "rb");
  RPDE_REQUIRE(fp != nullptr, "h5lite: cannot open " + filename);
  f_ = fp;
  struct Closer {   // a throwing constructor never reaches ~Reader
    void** f; bool armed = true;
    ~Closer() { if (armed && *f) { std::fclose(static_cast<FILE*>(*f)); *f = nullptr; } }
  } closer{&f_};
  std::fseek(fp, 0, SEEK_END);
  size_ = (uint64_t)std::ftell(fp);
  uint8_t sb[128] = {0};
  RPDE_REQUIRE(size_ >= 96, "h5lite: not an HDF5 file (too short): " + filename);
  pread_(sb, 0, 96 + 4);
  RPDE_REQUIRE(std::memcmp(sb, kSig, 8) == 0, "h5lite: not an HDF5 file (signature at offset 0): " + filename);
  const int ver = sb[8];
  RPDE_REQUIRE(ver == 0 || ver == 1,
               "h5lite: superblock version " + std::to_string(ver) + " (only the classic versions 0 / 1 are read)");
  RPDE_REQUIRE(sb[13] == 8 && sb[14] == 8, "h5lite: 8-byte offsets and lengths expected");
  leaf_k_ = (int)le(sb + 16, 2);
  internal_k_ = (int)le(sb + 18, 2);
  const int o = (ver == 1) ? 4 : 0;   // version 1 inserts the indexed-storage K and 2 reserved bytes
  base_ = le(sb + 24 + o, 8);
  const uint64_t root_oh = le(sb + 56 + o + 8, 8);
  walk_group(base_ + root_oh, "", 0);
  closer.armed = false;
}

Reader::~Reader() { if (f_) std::fclose(static_cast<FILE*>(f_)); }

void Reader::pread_(void* dst, uint64_t off, uint64_t n) const {
  RPDE_REQUIRE(n <= size_ && off <= size_ - n, "h5lite: read past the end of the file (corrupt address)");
  FILE* fp = static_cast<FILE*>(f_);
  std::fseek(fp, (long)off, SEEK_SET);
  RPDE_REQUIRE(std::fread(dst, 1, n, fp) == n, "h5lite: short read");
}

std::vector<Reader::Msg> Reader::object_messages(uint64_t oh) const {
  uint8_t h[16];
  pread_(h, oh, 16);
  RPDE_REQUIRE(h[0] == 1, "h5lite: object header version " + std::to_string(h[0]) + " (only version 1 is read)");
  int left = (int)le(h + 2, 2);
  std::vector<std::pair<uint64_t, uint64_t>> chunks{{oh + 16, le(h + 8, 4)}};
  std::vector<Msg> out;
  for (size_t c = 0; c < chunks.size() && left > 0; ++c) {
    std::vector<uint8_t> buf(chunks[c].second);
    pread_(buf.data(), chunks[c].first, buf.size());
    size_t p = 0;
    while (p + 8 <= buf.size() && left > 0) {
      const uint16_t type = (uint16_t)le(&buf[p], 2);
      const size_t sz = (size_t)le(&buf[p + 2], 2);
      RPDE_REQUIRE(p + 8 + sz <= buf.size(), "h5lite: header message overruns its chunk");
      --left;
      if (type == 0x0010) chunks.push_back({base_ + le(&buf[p + 8], 8), le(&buf[p + 16], 8)});   // continuation
      else if (type != 0) out.push_back(Msg{type, std::vector<uint8_t>(buf.begin() + p + 8, buf.begin() + p + 8 + sz)});
      p += 8 + sz;
    }
  }
  return out;
}

std::string Reader::heap_string(uint64_t heap_data, uint64_t off) const {
  std::string s;
  for (uint64_t p = heap_data + off;; ++p) {
    char c;
    pread_(&c, p, 1);
    if (!c) break;
    s.push_back(c);
    RPDE_REQUIRE(s.size() < 4096, "h5lite: unterminated name in a local heap");
  }
  return s;
}

void Reader::walk_btree(uint64_t node, uint64_t heap_data, const std::string& prefix, int depth) {
  uint8_t h[24];
  pread_(h, node, 24);
  RPDE_REQUIRE(std::memcmp(h, "TREE", 4) == 0 && h[4] == 0, "h5lite: group B-tree node expected");
  const int level = h[5], used = (int)le(h + 6, 2);
  for (int i = 0; i < used; ++i) {
    uint8_t c[8];
    pread_(c, node + 24 + 8 + (uint64_t)i * 16, 8);   // key i, CHILD i, key i+1, ...
    const uint64_t child = base_ + le(c, 8);
    if (level > 0) { walk_btree(child, heap_data, prefix, depth); continue; }
    uint8_t s[8];
    pread_(s, child, 8);
    RPDE_REQUIRE(std::memcmp(s, "SNOD", 4) == 0, "h5lite: symbol table node expected");
    const int nsym = (int)le(s + 6, 2);
    for (int k = 0; k < nsym; ++k) {
      uint8_t e[40];
      pread_(e, child + 8 + (uint64_t)k * 40, 40);
      const std::string name = heap_string(heap_data, le(e, 8));
      parse_object(base_ + le(e + 8, 8), prefix.empty() ? name : prefix + "/" + name, depth);
    }
  }
}

void Reader::walk_group(uint64_t oh, const std::string& prefix, int depth) {
  RPDE_REQUIRE(depth <= 4, "h5lite: groups nested too deep");
  for (const Msg& m : object_messages(oh))
    if (m.type == 0x0011) {
      RPDE_REQUIRE(m.data.size() >= 16, "h5lite: short symbol table message");
      uint8_t hp[32];
      pread_(hp, base_ + le(&m.data[8], 8), 32);
      RPDE_REQUIRE(std::memcmp(hp, "HEAP", 4) == 0, "h5lite: local heap expected");
      walk_btree(base_ + le(&m.data[0], 8), base_ + le(hp + 24, 8), prefix, depth);
      return;
    }
  fail("h5lite: group without a symbol table message (new-style groups are not read)");
}

void Reader::parse_object(uint64_t oh, const std::string& path, int depth) {
  const std::vector<Msg> msgs = object_messages(oh);
  bool group = false, has_space = false, has_layout = false, f64 = false;
  Entry e;
  for (const Msg& m : msgs) {
    const std::vector<uint8_t>& d = m.data;
    if (m.type == 0x0011) group = true;
    if (m.type == 0x0001) {   // dataspace
      RPDE_REQUIRE(d.size() >= 4, "h5lite: short dataspace message");
      const int ver = d[0], rank = d[1];
      const size_t off = ver == 1 ? 8 : 4;
      RPDE_REQUIRE((ver == 1 || ver == 2) && d.size() >= off + (size_t)rank * 8, "h5lite: dataspace version / size");
      for (int r = 0; r < rank; ++r) e.dims.push_back(le(&d[off + (size_t)r * 8], 8));
      has_space = true;
    }
    if (m.type == 0x0003) {   // datatype: IEEE f64 little endian, or a 64-bit little-endian integer (scalars such as num_save)
      RPDE_REQUIRE(d.size() >= 8, "h5lite: short datatype message");
      f64 = (d[0] & 0x0f) == 1 && (d[1] & 1) == 0 && le(&d[4], 4) == 8;
      if ((d[0] & 0x0f) == 0 && (d[1] & 1) == 0 && le(&d[4], 4) == 8) { f64 = true; e.kind = (d[1] & 0x08) ? 2 : 1; }
    }
    if (m.type == 0x0008) {   // layout
      RPDE_REQUIRE(d.size() >= 2, "h5lite: short layout message");
      RPDE_REQUIRE(d[0] == 3, "h5lite: data layout version " + std::to_string(d[0]) + " of " + path + " (version 3 is read)");
      if (d[1] == 1) { e.addr = le(&d[2], 8); e.bytes = le(&d[10], 8); }
      else if (d[1] == 0) { e.compact = true; const size_t n = (size_t)le(&d[2], 2); e.inline_data.assign(d.begin() + 4, d.begin() + 4 + n); e.bytes = n; }
      else fail("h5lite: dataset " + path + " is chunked; the snapshots are written with no_chunk()");
      has_layout = true;
    }
  }
  if (group) { walk_group(oh, path, depth + 1); return; }
  if (!(has_space && has_layout)) return;   // not a dataset we understand (e.g. a named datatype)
  RPDE_REQUIRE(f64, "h5lite: dataset " + path + " is not little-endian f64");
  index_[path] = std::move(e);
}

std::vector<std::string> Reader::paths() const {
  std::vector<std::string> p;
  for (const auto& kv : index_) p.push_back(kv.first);
  return p;
}

std::vector<uint64_t> Reader::shape(const std::string& path) const {
  auto it = index_.find(path);
  RPDE_REQUIRE(it != index_.end(), "h5lite: no dataset \"" + path + "\"");
  return it->second.dims;
}

Dataset Reader::read(const std::string& path) const {
  auto it = index_.find(path);
  RPDE_REQUIRE(it != index_.end(), "h5lite: no dataset \"" + path + "\"");
  const Entry& e = it->second;
  Dataset d;
  d.dims = e.dims;
  uint64_t n = 1;
  for (uint64_t v : e.dims) n *= v;
  d.data.assign(n, 0.0);
  if (e.compact) {
    RPDE_REQUIRE(e.inline_data.size() == n * 8, "h5lite: compact dataset size");
    std::memcpy(d.data.data(), e.inline_data.data(), n * 8);
  } else if (e.addr != kUndef && n) {   // an undefined address = never written: zeros
    RPDE_REQUIRE(e.bytes == n * 8, "h5lite: contiguous dataset size");
    pread_(d.data.data(), base_ + e.addr, n * 8);
  }
  if (e.kind != 0) {   // integers on disk: convert in place
    d.u64 = e.kind == 1;
    for (uint64_t i = 0; i < n; ++i) {
      uint64_t raw;
      std::memcpy(&raw, &d.data[i], 8);
      d.data[i] = e.kind == 1 ? (double)raw : (double)(int64_t)raw;
    }
  }
  return d;
}

}  // namespace h5
}  // namespace rpde
