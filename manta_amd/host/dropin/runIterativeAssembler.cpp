// Replaces Manta's assembly/IterativeAssembler.cpp at link time (see manta_amd_dropin.hpp): the reference's own entry point
//   void runIterativeAssembler(const IterativeAssemblerOptions&, AssemblyReadInput&, AssemblyReadOutput&, Assembly&)
//                                                                      assembly/IterativeAssembler.hpp:43-47
// on manta_assemble_batch, written against Manta's real headers.
#include "assembly/IterativeAssembler.hpp"

#include <string>
#include <vector>

#include "manta_amd_dropin.hpp"

void runIterativeAssembler(const IterativeAssemblerOptions& opt, AssemblyReadInput& reads, AssemblyReadOutput& assembledReadInfo, Assembly& contigs)
{
  using illumina::common::GeneralException;
  manta_ctx_t*        ctx = manta_amd_dropin::threadContext();
  manta_asm_options_t o;
  o.min_word_length           = opt.minWordLength;
  o.max_word_length           = opt.maxWordLength;
  o.word_step_size            = opt.wordStepSize;
  o.min_contig_length         = opt.minContigLength;
  o.min_coverage              = opt.minCoverage;
  o.min_conservative_coverage = opt.minConservativeCoverage;
  o.min_unused_reads          = opt.minUnusedReads;
  o.min_support_reads         = opt.minSupportReads;
  o.max_assembly_count        = opt.maxAssemblyCount;
  const unsigned        nReads = unsigned(reads.size());
  std::vector<uint8_t>  bases;
  std::vector<uint64_t> readOff(nReads + 1, 0);
  for (unsigned r = 0; r < nReads; ++r) {
    bases.insert(bases.end(), reads[r].begin(), reads[r].end());
    readOff[r + 1] = bases.size();
  }
  bases.push_back(0);
  const uint32_t                  locusBegin[2] = {0, nReads};
  manta_asm_locus_result_t        locus;
  std::vector<manta_asm_contig_t> recs(o.max_assembly_count + 1);
  uint64_t                        maxRead = 0;
  for (unsigned r = 0; r < nReads; ++r) maxRead = std::max<uint64_t>(maxRead, reads[r].size());
  std::vector<uint8_t>  seq(size_t(3 * o.max_assembly_count + 2) * (bases.size() + opt.maxWordLength + 64) + 65536);
  std::vector<uint64_t> bits(size_t(o.max_assembly_count) * 2 * ((nReads + 2 * o.max_assembly_count + 63) / 64 + 1) + 4 * o.max_assembly_count + 64);
  uint64_t              seqUsed = 0, bitsUsed = 0;
  const int rc = manta_assemble_batch(ctx, &o, 1, bases.data(), readOff.data(), locusBegin, &locus, recs.data(), recs.size(), seq.data(), seq.size(),
                                      &seqUsed, bits.data(), bits.size(), &bitsUsed);
  if (rc != MANTA_OK) BOOST_THROW_EXCEPTION(GeneralException(std::string("manta_amd assembler: ") + manta_last_error(ctx)));
  // contigs (IterativeAssembler.cpp:659, 818-841) and the pseudo reads the reference leaves appended to `reads` (:902)
  contigs.clear();
  contigs.resize(locus.n_contigs);
  for (uint32_t c = 0; c < locus.n_contigs; ++c) {
    const manta_asm_contig_t& rec(recs[locus.first_contig + c]);
    AssembledContig&          ctg(contigs[c]);
    ctg.seq.assign(reinterpret_cast<const char*>(seq.data() + rec.seq_off), rec.seq_len);
    ctg.seedReadCount = rec.seed_read_count;
    for (uint32_t w = 0; w < locus.n_words; ++w) {
      for (uint64_t s = bits[rec.support_off + w]; s; s &= s - 1) ctg.supportReads.insert(unsigned(64 * w + __builtin_ctzll(s)));
      for (uint64_t s = bits[rec.reject_off + w]; s; s &= s - 1) ctg.rejectReads.insert(unsigned(64 * w + __builtin_ctzll(s)));
    }
    ctg.conservativeRange.set_range(rec.conservative_begin, rec.conservative_end);
  }
  uint64_t off = locus.pseudo_seq_off;
  for (uint32_t p = 0; p < locus.n_pseudo; ++p) {
    const uint64_t len = bits[locus.pseudo_len_off + p];
    reads.push_back(std::string(reinterpret_cast<const char*>(seq.data() + off), len));
    off += len;
  }
  // readInfo (:858-870, 826-834): isUsed / contigIds from the contigs' support sets, isPseudo for the appended reads
  assembledReadInfo.clear();
  assembledReadInfo.resize(reads.size());
  for (unsigned r = nReads; r < reads.size(); ++r) assembledReadInfo[r].isPseudo = true;
  for (uint32_t c = 0; c < locus.n_contigs; ++c) {
    for (const unsigned r : contigs[c].supportReads) {
      if (r >= assembledReadInfo.size()) continue;  // stale pseudo-read index (the reference reads out of bounds there)
      assembledReadInfo[r].isUsed = true;
      assembledReadInfo[r].contigIds.push_back(c);
    }
  }
}
