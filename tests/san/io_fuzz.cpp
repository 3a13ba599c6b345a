// io_fuzz.cpp -- truncated and bit-flipped PNG / YAML files through the parsers of svo_hip::io (svo_hip_io.cpp) under
// AddressSanitizer + UBSan.  A damaged file may be refused (std::runtime_error) or parsed into something; it must never
// read or write outside its buffers, overflow an integer it trusts, or fall over.  Seeds: files written by the caller
// (tests/test_sanitizers_cpu.py: PNGs of several kinds, calibration and parameter YAMLs).
//   io_fuzz_asan png|yaml <seed file> <n_mutations> <rng seed>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <vector>

#include <zlib.h>

#include "../../svo_pro_universal_amd/host/svo_hip_io.h"

using namespace svo_hip;

static unsigned long long g_state = 1;
static unsigned rnd() { g_state = g_state * 6364136223846793005ULL + 1442695040888963407ULL; return (unsigned)(g_state >> 33); }

// ---- PNG-aware mutation: a flipped bit anywhere in a PNG is caught by a chunk CRC or by zlib long before the decoder's own
// arithmetic sees it.  To reach that arithmetic the file is taken apart (chunks; IDAT inflated), damaged INSIDE -- header
// fields, filter bytes, scanline bytes, the amount of data -- and put together again with correct CRCs and a valid stream.
static unsigned long be32(const unsigned char* p) { return ((unsigned long)p[0] << 24) | ((unsigned long)p[1] << 16) | ((unsigned long)p[2] << 8) | p[3]; }
static void put32(std::vector<unsigned char>& v, unsigned long x) { for (int s = 24; s >= 0; s -= 8) v.push_back((unsigned char)(x >> s)); }
static void put_chunk(std::vector<unsigned char>& out, const char* type, const std::vector<unsigned char>& data)
{
  put32(out, (unsigned long)data.size());
  std::vector<unsigned char> td(type, type + 4);
  td.insert(td.end(), data.begin(), data.end());
  out.insert(out.end(), td.begin(), td.end());
  put32(out, crc32(0L, td.data(), (uInt)td.size()));
}
static bool mutate_png_inside(const std::vector<unsigned char>& seed, std::vector<unsigned char>* out)
{
  if (seed.size() < 8 + 25) return false;
  std::vector<unsigned char> ihdr, idat;
  size_t at = 8;
  while (at + 12 <= seed.size()) {
    const unsigned long len = be32(&seed[at]);
    if (at + 12 + len > seed.size()) return false;
    const std::string type(seed.begin() + (long)at + 4, seed.begin() + (long)at + 8);
    if (type == "IHDR") ihdr.assign(seed.begin() + (long)at + 8, seed.begin() + (long)(at + 8 + len));
    if (type == "IDAT") idat.insert(idat.end(), seed.begin() + (long)at + 8, seed.begin() + (long)(at + 8 + len));
    at += 12 + len;
  }
  if (ihdr.size() != 13 || idat.empty()) return false;
  std::vector<unsigned char> raw(1 << 20);
  uLongf raw_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK) return false;
  raw.resize(raw_len);
  switch (rnd() % 6) {
    case 0: { const unsigned long v[] = { 0, 1, 7, 0x7fffffffUL, 0xffffffffUL, 65536, 3 }; const unsigned long x = v[rnd() % 7]; unsigned char* p = &ihdr[(rnd() & 1) * 4]; p[0] = (unsigned char)(x >> 24); p[1] = (unsigned char)(x >> 16); p[2] = (unsigned char)(x >> 8); p[3] = (unsigned char)x; break; }   // width / height
    case 1: ihdr[8 + rnd() % 5] = (unsigned char)(rnd() % 20); break;                        // bit depth, colour type, compression, filter, interlace
    case 2: for (int k = 0; k < 8 && !raw.empty(); ++k) raw[rnd() % raw.size()] = (unsigned char)rnd(); break;   // scanline bytes (filter bytes among them)
    case 3: if (!raw.empty()) raw.resize(rnd() % raw.size()); break;                        // less data than the header promises
    case 4: raw.resize(raw.size() + 1 + rnd() % 300, (unsigned char)rnd()); break;          // more
    default: { const unsigned long w = be32(&ihdr[0]); if (w) for (size_t r = 0; r < raw.size(); r += (size_t)(w * (raw.size() / (be32(&ihdr[4]) ? be32(&ihdr[4]) : 1) / w ? raw.size() / (be32(&ihdr[4]) ? be32(&ihdr[4]) : 1) / w : 1)) + 1) raw[r] = (unsigned char)(rnd() % 7); break; }   // filter types, valid and not
  }
  std::vector<unsigned char> comp(compressBound((uLong)raw.size()) + 16);
  uLongf comp_len = (uLongf)comp.size();
  if (compress2(comp.data(), &comp_len, raw.data(), (uLong)raw.size(), 1) != Z_OK) return false;
  comp.resize(comp_len);
  out->assign(seed.begin(), seed.begin() + 8);
  put_chunk(*out, "IHDR", ihdr);
  if (rnd() % 4 == 0 && comp.size() > 2) {   // IDAT in two pieces
    const size_t cut = 1 + rnd() % (comp.size() - 1);
    put_chunk(*out, "IDAT", std::vector<unsigned char>(comp.begin(), comp.begin() + (long)cut));
    put_chunk(*out, "IDAT", std::vector<unsigned char>(comp.begin() + (long)cut, comp.end()));
  } else put_chunk(*out, "IDAT", comp);
  put_chunk(*out, "IEND", std::vector<unsigned char>());
  return true;
}

int main(int argc, char** argv)
{
  if (argc < 5) return 2;
  const std::string what = argv[1];
  std::ifstream f(argv[2], std::ios::binary);
  const std::vector<unsigned char> seed((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (seed.empty()) { fprintf(stderr, "empty seed file\n"); return 2; }
  const int n = atoi(argv[3]);
  g_state = (unsigned long long)atoll(argv[4]) * 2654435761ULL + 1;
  long parsed = 0, refused = 0;
  auto feed = [&](const std::vector<unsigned char>& bytes) {
    try {
      if (what == "png") {
        const io::GrayImage img = io::decodePngGray(bytes.data(), bytes.size());
        if (img.data.size() != (size_t)img.width * img.height) { fprintf(stderr, "image size and data disagree\n"); abort(); }
      } else {
        const io::YamlNode root = io::parseYaml(std::string(bytes.begin(), bytes.end()));
        // what the loaders do with a parsed tree: every accessor, the rig, the parameters
        try { (void)io::cameraRigFromYaml(root); } catch (const std::runtime_error&) {}
        try { (void)io::frontendParamsFromYaml(root); } catch (const std::runtime_error&) {}
        if (root["cameras"].kind == io::YamlNode::kSeq && root["cameras"].size() > 0) (void)root["cameras"][0]["camera"]["intrinsics"]["data"].asDoubles();
      }
      ++parsed;
    } catch (const std::runtime_error&) { ++refused; }
      catch (const std::out_of_range&) { ++refused; }
      catch (const std::invalid_argument&) { ++refused; }
      catch (const std::bad_alloc&) { ++refused; }   // a length field that asks for the moon is refused by the allocator at worst
  };
  feed(seed);
  // every prefix length on a coarse grid, then the tail lengths one by one
  for (size_t len = 0; len < seed.size(); len += 1 + seed.size() / 97) feed(std::vector<unsigned char>(seed.begin(), seed.begin() + (long)len));
  for (size_t cut = 1; cut <= 24 && cut < seed.size(); ++cut) feed(std::vector<unsigned char>(seed.begin(), seed.end() - (long)cut));
  for (int k = 0; k < n; ++k) {
    std::vector<unsigned char> m = seed;
    const int flips = 1 + (int)(rnd() % 4);
    for (int j = 0; j < flips; ++j) {
      const size_t at = rnd() % m.size();
      switch (rnd() % 4) {
        case 0: m[at] ^= (unsigned char)(1u << (rnd() % 8)); break;        // one bit
        case 1: m[at] = (unsigned char)rnd(); break;                       // one byte
        case 2: m[at] = (rnd() & 1) ? 0xFF : 0x00; break;                  // an extreme
        default: if (at + 4 <= m.size()) for (int b = 0; b < 4; ++b) m[at + (size_t)b] = (unsigned char)rnd(); break;   // a length / CRC field
      }
    }
    if (rnd() % 5 == 0) m.resize(rnd() % (m.size() + 1));
    feed(m);
  }
  if (what == "png")
    for (int k = 0; k < n; ++k) {
      std::vector<unsigned char> m;
      if (mutate_png_inside(seed, &m)) feed(m);
    }
  printf("%s: %ld parsed, %ld refused\n", what.c_str(), parsed, refused);
  return 0;
}
