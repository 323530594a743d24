// Test-only C surface over manta_amd/host/read_gather.hpp (ReadGatherBatch incl. the remote-mate retrieval): the Python tests feed
// it the records of region queries (the reference's own BAM layer, oracle/_ref/libmanta_ref_bam.so, plays the caller's
// bam_streamer) and read the final piles back as text.
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "read_gather.hpp"

using namespace manta_amd;
#define MINE_EXPORT extern "C" __attribute__((visibility("default")))

typedef void (*rg_scan_cb)(void* user, uint32_t bamIndex, int32_t tid, int32_t begin, int32_t end);

namespace {
struct Handle : RemoteMateFetcher {
  ReadGatherBatch                                  batch;
  rg_scan_cb                                       cb   = nullptr;
  void*                                            user = nullptr;
  const std::function<bool(const RemoteRecord&)>*  sink = nullptr;
  bool                                             sinkOpen = false;
  unsigned                                         nQueries = 0;
  std::string                                      error;
  void scanRegion(uint32_t bamIndex, int32_t tid, int32_t begin, int32_t end, const std::function<bool(const RemoteRecord&)>& s) override
  {
    sink     = &s;
    sinkOpen = true;
    ++nQueries;
    cb(user, bamIndex, tid, begin, end);  // the Python side calls rg_remote_record for every record of the region
    sink = nullptr;
  }
};
int emit(const std::string& s, char* out, const int cap)
{
  if (int(s.size()) + 1 > cap) return -int(s.size()) - 1;
  std::memcpy(out, s.c_str(), s.size() + 1);
  return int(s.size());
}
}  // namespace

MINE_EXPORT void* rg_new() { return new Handle(); }
MINE_EXPORT void  rg_free(void* h) { delete static_cast<Handle*>(h); }
MINE_EXPORT void  rg_begin_candidate(void* h, int isMaxDepth, float maxDepth, float maxLocal, int searchRemote)
{
  static_cast<Handle*>(h)->batch.beginCandidate(isMaxDepth != 0, maxDepth, maxLocal, searchRemote != 0);
}
MINE_EXPORT void rg_begin_query(
    void* h, int bpBegin, int bpEnd, int bpState, int isLocusReversed, unsigned bamIndex, int isTumor, int first, int refOffset, const char* refSeq)
{
  static_cast<Handle*>(h)->batch.beginQuery(bpBegin, bpEnd, bpState, isLocusReversed != 0, bamIndex, isTumor != 0, first != 0, refOffset, refSeq);
}
MINE_EXPORT void rg_add_record(
    void* h, int tid, int pos, int mtid, int mpos, unsigned flag, unsigned mapq, const uint32_t* cigar, unsigned nCigar, const char* qname,
    const uint8_t* seq4, const uint8_t* qual, unsigned lQseq, int hasSA, const char* mateCigar)
{
  static_cast<Handle*>(h)->batch.addRecord(tid, pos, mtid, mpos, uint16_t(flag), uint8_t(mapq), cigar, nCigar, qname, seq4, qual, lQseq, hasSA != 0,
                                           mateCigar);
}
/// run on the calling thread's context (manta_amd.hpp threadContext()); 0 or -1 (rg_error)
MINE_EXPORT int rg_run(void* hv, const manta_read_class_options_t* opt)
{
  Handle* h = static_cast<Handle*>(hv);
  try {
    h->batch.run(threadContext(), *opt);
    return 0;
  } catch (const std::exception& e) {
    h->error = e.what();
    return -1;
  }
}
MINE_EXPORT int rg_retrieve_remote(void* hv, const manta_read_class_options_t* opt, rg_scan_cb cb, void* user)
{
  Handle* h = static_cast<Handle*>(hv);
  h->cb     = cb;
  h->user   = user;
  try {
    h->batch.retrieveRemoteReads(*h, *opt);
    return int(h->nQueries);
  } catch (const std::exception& e) {
    h->error = e.what();
    return -1;
  }
}
/// one record of the region query in progress; returns 0 once the scan wants no more records
MINE_EXPORT int rg_remote_record(void* hv, int pos, unsigned flag, unsigned mapq, const char* qname, const uint8_t* seq4, const uint8_t* qual,
                                 unsigned lQseq, int hasSA)
{
  Handle* h = static_cast<Handle*>(hv);
  if (!h->sink || !h->sinkOpen) return 0;
  RemoteRecord r;
  r.pos   = pos;
  r.flag  = uint16_t(flag);
  r.mapq  = uint8_t(mapq);
  r.qname = qname;
  r.seq4  = seq4;
  r.qual  = qual;
  r.lQseq = lQseq;
  r.hasSA = hasSA != 0;
  h->sinkOpen = (*h->sink)(r);
  return h->sinkOpen ? 1 : 0;
}
MINE_EXPORT int rg_error(void* h, char* out, int cap) { return emit(static_cast<Handle*>(h)->error, out, cap); }
/// "status <s> reads <n>" + the final pile of candidate l, one read per line
MINE_EXPORT int rg_pile_text(void* hv, unsigned l, char* out, int cap)
{
  Handle*           h = static_cast<Handle*>(hv);
  const uint32_t    b = h->batch.finalLocusReadBegin(l), e = h->batch.finalLocusReadBegin(l + 1);
  std::string       s = "status " + std::to_string(h->batch.results[l].status) + " reads " + std::to_string(e - b) + "\n";
  for (uint32_t r = b; r < e; ++r) s += h->batch.finalPileReadText(r) + "\n";
  return emit(s, out, cap);
}
/// the RemoteReadCache entries of candidate l, by name: "<qname> <readNo> <read>"
MINE_EXPORT int rg_remote_cache(void* hv, unsigned l, char* out, int cap)
{
  Handle*                  h = static_cast<Handle*>(hv);
  std::map<std::string, std::string> byName;  // (the reference's cache is a map by name: a later entry of the same name replaces the earlier)
  for (const RemoteReadCacheEntry& c : h->batch.remoteCache)
    if (c.candidate == l) byName[c.qname] = std::to_string(c.readNo) + " " + h->batch.finalPileReadText(c.pileRead);
  std::string s;
  for (const auto& kv : byName) s += kv.first + " " + kv.second + "\n";
  return emit(s, out, cap);
}
MINE_EXPORT unsigned long long rg_counts(void* hv, int which)
{
  Handle* h = static_cast<Handle*>(hv);
  return which == 0 ? h->batch.nRemoteTargets : h->batch.nRemoteInserted;
}
