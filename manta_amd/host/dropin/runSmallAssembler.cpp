// Replaces Manta's assembly/SmallAssembler.cpp at link time (see manta_amd_dropin.hpp): the reference's own entry point
//   void runSmallAssembler(const SmallAssemblerOptions&, const AssemblyReadInput&, AssemblyReadOutput&, Assembly&)
//                                                                      assembly/SmallAssembler.hpp:43-47
// on manta_small_assemble_batch, written against Manta's real headers.
#include "assembly/SmallAssembler.hpp"

#include <string>
#include <vector>

#include "manta_amd_dropin.hpp"

void runSmallAssembler(
    const SmallAssemblerOptions& opt, const AssemblyReadInput& reads, AssemblyReadOutput& assembledReadInfo, Assembly& contigs)
{
  using illumina::common::GeneralException;
  if (opt.alphabet != "ACGT") BOOST_THROW_EXCEPTION(GeneralException("manta_amd assembler: only the default alphabet \"ACGT\" is supported"));
  manta_ctx_t*              ctx = manta_amd_dropin::threadContext();
  manta_small_asm_options_t o;
  o.min_word_length           = opt.minWordLength;
  o.max_word_length           = opt.maxWordLength;
  o.word_step_size            = opt.wordStepSize;
  o.min_contig_length         = opt.minContigLength;
  o.min_coverage              = opt.minCoverage;
  o.min_conservative_coverage = opt.minConservativeCoverage;
  o.min_seed_reads            = opt.minSeedReads;
  o.max_assembly_iterations   = opt.maxAssemblyIterations;
  const unsigned        nReads = unsigned(reads.size());
  std::vector<uint8_t>  bases;
  std::vector<uint64_t> readOff(nReads + 1, 0);
  for (unsigned r = 0; r < nReads; ++r) {
    bases.insert(bases.end(), reads[r].begin(), reads[r].end());
    readOff[r + 1] = bases.size();
  }
  bases.push_back(0);
  const uint32_t                  locusBegin[2] = {0, nReads};
  const unsigned                  nSlots        = o.max_assembly_iterations + 1;  // contigs + the isFiltered record
  manta_asm_locus_result_t        locus;
  std::vector<manta_asm_contig_t> recs(nSlots + 1);
  std::vector<uint8_t>            seq(size_t(nSlots + 1) * (bases.size() + opt.maxWordLength + 64) + 65536);
  std::vector<uint64_t>           bits(size_t(nSlots) * 2 * ((nReads + 63) / 64 + 1) + 64);
  uint64_t                        seqUsed = 0, bitsUsed = 0;
  const int rc = manta_small_assemble_batch(ctx, &o, 1, bases.data(), readOff.data(), locusBegin, &locus, recs.data(), recs.size(), seq.data(),
                                            seq.size(), &seqUsed, bits.data(), bits.size(), &bitsUsed);
  if (rc != MANTA_OK) BOOST_THROW_EXCEPTION(GeneralException(std::string("manta_amd assembler: ") + manta_last_error(ctx)));
  assembledReadInfo.clear();
  contigs.clear();
  assembledReadInfo.resize(nReads);
  for (uint32_t c = 0; c < locus.n_contigs; ++c) {
    const manta_asm_contig_t& rec(recs[locus.first_contig + c]);
    if (rec.seed_read_count == 0xffffffffu) {  // reads dropped for holding a word twice (SmallAssembler.cpp:496-503)
      for (uint32_t w = 0; w < locus.n_words; ++w)
        for (uint64_t s = bits[rec.support_off + w]; s; s &= s - 1) {
          AssemblyReadInfo& rinfo(assembledReadInfo[64 * w + __builtin_ctzll(s)]);
          rinfo.isUsed     = true;
          rinfo.isFiltered = true;
        }
      continue;
    }
    contigs.emplace_back();
    AssembledContig& ctg(contigs.back());
    ctg.seq.assign(reinterpret_cast<const char*>(seq.data() + rec.seq_off), rec.seq_len);
    ctg.seedReadCount = rec.seed_read_count;
    for (uint32_t w = 0; w < locus.n_words; ++w) {
      for (uint64_t s = bits[rec.support_off + w]; s; s &= s - 1) ctg.supportReads.insert(unsigned(64 * w + __builtin_ctzll(s)));
      for (uint64_t s = bits[rec.reject_off + w]; s; s &= s - 1) ctg.rejectReads.insert(unsigned(64 * w + __builtin_ctzll(s)));
    }
    ctg.conservativeRange.set_range(rec.conservative_begin, rec.conservative_end);
    for (const unsigned r : ctg.supportReads) {  // :594-606
      assembledReadInfo[r].isUsed = true;
      assembledReadInfo[r].contigIds.push_back(unsigned(contigs.size() - 1));
    }
  }
}
