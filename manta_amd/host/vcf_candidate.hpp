// candidateSV.vcf records of refined candidates (record formation only): the host-side step
// that turns what the refiner produced into the text Manta's candidate VCF holds.  Restates
//   format/VcfWriterSV.cpp:247-698 (writeTransloc / writeIndel / writeSVCore, addHomologyInfo, makeInfoField),
//   format/VcfWriterCandidateSV.cpp:34-66 (the PAIR_COUNT tags), manta/JunctionIdGenerator.{hpp,cpp} (record ids)
// (paths relative to /root/reference/src/c++/lib) over sv_types.hpp.  Reference bases (REF, HOMSEQ) come through the
// same RefinerInputSource callback the refiner uses.  Pinned against the reference's own writer objects, compiled
// unmodified (oracle/ref_refiner_driver.cpp: ref_candidate_vcf_records) by tests/test_vcf_candidate.py.
#pragma once

#include <ostream>
#include <sstream>

#include "refiner.hpp"

namespace manta_amd {

/// svgraph/EdgeInfo.hpp:30-39
struct EdgeInfo {
  unsigned locusIndex = 0, nodeIndex1 = 0, nodeIndex2 = 0;
};

/// manta/JunctionIdGenerator.hpp:35-41
struct SVId {
  const char*               getLabel() const { return EXTENDED_SV_TYPE::label(svType); }
  EXTENDED_SV_TYPE::index_t svType = EXTENDED_SV_TYPE::UNKNOWN;
  std::string               localId, mateId;
};

/// manta/JunctionIdGenerator.cpp:25-43
struct JunctionIdGenerator {
  void getId(const EdgeInfo& edge, const SVCandidate& sv, const bool isRNA, SVId& svId) const
  {
    svId.svType = getExtendedSVType(sv, isRNA);
    std::ostringstream os;
    os << "Manta" << EXTENDED_SV_TYPE::label(svId.svType) << ':' << edge.locusIndex << ':' << edge.nodeIndex1 << ':' << edge.nodeIndex2 << ':'
       << sv.candidateIndex << ':' << sv.assemblyAlignIndex << ':' << sv.assemblySegmentIndex;
    svId.localId = os.str();
    if (EXTENDED_SV_TYPE::isSVTransloc(svId.svType) || EXTENDED_SV_TYPE::isSVInv(svId.svType)) {
      svId.mateId  = svId.localId + ":1";
      svId.localId = svId.localId + ":0";
    } else {
      svId.mateId.clear();
    }
  }
};

struct VcfWriterCandidateSV {
  VcfWriterCandidateSV(RefinerInputSource& source, const bam_header_info& header, std::ostream& os, const bool isOutputContig = false)
    : _source(source), _header(header), _os(os), _isOutputContig(isOutputContig)
  {
  }

  /// VcfWriterCandidateSV::writeSV (VcfWriterCandidateSV.cpp:58-66) -> VcfWriterSV::writeSVCore (VcfWriterSV.cpp:651-686)
  void writeSV(const SVCandidate& sv, const SVId& svId) const
  {
    using namespace EXTENDED_SV_TYPE;
    const index_t svType(getExtendedSVType(sv));
    if (svType == UNKNOWN) throw GeneralException("SV candidate cannot be classified");
    if (isSVTransloc(svType) || isSVInv(svType)) {
      writeTransloc(sv, svId, true);
      writeTransloc(sv, svId, false);
    } else {
      writeIndel(sv, svId, isSVIndel(svType));
    }
  }

  /// VcfWriterSV::writeHeader (format/VcfWriterSV.cpp:58-131) with VcfWriterCandidateSV::addHeaderInfo
  /// (format/VcfWriterCandidateSV.cpp:26-32): the header block of candidateSV.vcf.  `fileDate` is the reference's
  /// vcf_fileDate (format/VcfWriterSV.cpp:70: the run's date, one of the five header keys every Manta comparison skips,
  /// src/demo/runMantaWorkflowDemo.py `rexclude`); the text is fixed by the VCF output contract.
  void writeHeader(const char* progName, const char* progVersion, const std::string& referenceFilename, const std::string& fileDate,
                   const std::vector<std::string>& sampleNames = std::vector<std::string>()) const
  {
    std::ostream& os(_os);
    os << "##fileformat=VCFv4.1\n";
    os << "##fileDate=" << fileDate << "\n";
    os << "##source=" << progName << " " << progVersion << "\n";
    os << "##reference=file://" << referenceFilename << "\n";
    for (const bam_header_info::chrom_info& cdata : _header.chrom_data) os << "##contig=<ID=" << cdata.label << ",length=" << cdata.length << ">\n";
    static const char* const reservedInfo[][4] = {
        {"IMPRECISE", "0", "Flag", "Imprecise structural variation"},
        {"SVTYPE", "1", "String", "Type of structural variant"},
        {"SVLEN", ".", "Integer", "Difference in length between REF and ALT alleles"},
        {"END", "1", "Integer", "End position of the variant described in this record"},
        {"CIPOS", "2", "Integer", "Confidence interval around POS"},
        {"CIEND", "2", "Integer", "Confidence interval around END"},
        {"CIGAR", "A", "String", "CIGAR alignment for each alternate indel allele"},
        {"MATEID", ".", "String", "ID of mate breakend"},
        {"EVENT", "1", "String", "ID of event associated to breakend"},
        {"HOMLEN", ".", "Integer", "Length of base pair identical homology at event breakpoints"},
        {"HOMSEQ", ".", "String", "Sequence of base pair identical homology at event breakpoints"},
        {"SVINSLEN", ".", "Integer", "Length of insertion"},
        {"SVINSSEQ", ".", "String", "Sequence of insertion"},
        {"LEFT_SVINSSEQ", ".", "String", "Known left side of insertion for an insertion of unknown length"},
        {"RIGHT_SVINSSEQ", ".", "String", "Known right side of insertion for an insertion of unknown length"}};
    auto info = [&](const char* const* t) {
      os << "##INFO=<ID=" << t[0] << ",Number=" << t[1] << ",Type=" << t[2] << ",Description=\"" << t[3] << "\">\n";
    };
    for (const auto& t : reservedInfo) info(t);
    if (_isOutputContig) {
      static const char* const contig[4] = {"CONTIG", "1", "String", "Assembled contig sequence"};
      info(contig);
    }
    static const char* const candidateInfo[][4] = {
        {"PAIR_COUNT", "1", "Integer", "Read pairs supporting this variant where both reads are confidently mapped"},
        {"BND_PAIR_COUNT", "1", "Integer",
         "Confidently mapped reads supporting this variant at this breakend (mapping may not be confident at remote breakend)"},
        {"UPSTREAM_PAIR_COUNT", "1", "Integer",
         "Confidently mapped reads supporting this variant at the upstream breakend (mapping may not be confident at downstream breakend)"},
        {"DOWNSTREAM_PAIR_COUNT", "1", "Integer",
         "Confidently mapped reads supporting this variant at this downstream breakend (mapping may not be confident at upstream breakend)"}};
    for (const auto& t : candidateInfo) info(t);
    os << "##ALT=<ID=DEL,Description=\"Deletion\">\n";
    os << "##ALT=<ID=INS,Description=\"Insertion\">\n";
    os << "##ALT=<ID=DUP:TANDEM,Description=\"Tandem Duplication\">\n";
    os << "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO";
    if (!sampleNames.empty()) {
      os << "\tFORMAT";
      for (const std::string& sampleName : sampleNames) os << '\t' << sampleName;
    }
    os << '\n';
  }

private:
  typedef std::vector<std::string> InfoTag_t;

  static std::string num(const long v) { return std::to_string(v); }

  std::string refSeq(const std::string& chrom, const pos_t beginPos, const pos_t endPos) const
  {
    std::string s;
    _source.getReferenceSeq(chrom, beginPos, endPos, s);
    return s;
  }

  /// HOMLEN / HOMSEQ of a breakend homology range (:258-277)
  void addHomologyInfo(const std::string& chrom, const known_pos_range2& bpRange, const pos_t bpPosAdjust, InfoTag_t& infoTags) const
  {
    if (bpRange.size() <= 1) return;
    infoTags.push_back("HOMLEN=" + num(long(bpRange.size()) - 1));
    infoTags.push_back("HOMSEQ=" + refSeq(chrom, bpRange.begin_pos() + bpPosAdjust + 1, bpRange.end_pos() + bpPosAdjust - 1));
  }

  void emit(const std::string& chrom, const pos_t pos, const std::string& id, const std::string& ref, const std::string& alt,
            const InfoTag_t& infoTags) const
  {
    _os << chrom << '\t' << pos << '\t' << id << '\t' << ref << '\t' << alt << "\t.\t.\t";  // QUAL and FILTER are '.' for candidates
    for (size_t i = 0; i < infoTags.size(); ++i) _os << (i ? ";" : "") << infoTags[i];
    _os << '\n';
  }

  /// one breakend record of a translocation / inversion junction (:279-423)
  void writeTransloc(const SVCandidate& sv, const SVId& svId, const bool isFirstBreakend) const
  {
    const bool              isImprecise(sv.isImprecise());
    const SVBreakend&       bpA(isFirstBreakend ? sv.bp1 : sv.bp2);
    const SVBreakend&       bpB(isFirstBreakend ? sv.bp2 : sv.bp1);
    const std::string&      chrom(_header.chrom_data.at(size_t(bpA.interval.tid)).label);
    const std::string&      mateChrom(_header.chrom_data.at(size_t(bpB.interval.tid)).label);
    const known_pos_range2& bpARange(bpA.interval.range);
    const known_pos_range2& bpBRange(bpB.interval.range);
    pos_t                   pos     = bpARange.center_pos() + 1;
    pos_t                   matePos = bpBRange.center_pos() + 1;
    if (!isImprecise) {
      pos     = bpARange.begin_pos() + 1;
      matePos = sv.isBreakendRangeSameShift() ? (bpBRange.begin_pos() + 1) : bpBRange.end_pos();
    }
    if (pos < 1 || matePos < 1) return;
    const std::string& localId(isFirstBreakend ? svId.localId : svId.mateId);
    const std::string& mateId(isFirstBreakend ? svId.mateId : svId.localId);
    const std::string  ref(refSeq(chrom, pos - 1, pos - 1));
    std::string        insertSeq(sv.insertSeq);
    if (!(isFirstBreakend || (bpA.state != bpB.state))) reverseCompStr(insertSeq);

    std::string altPrefix, altSuffix;
    if (bpA.state == SVBreakendState::RIGHT_OPEN)
      altPrefix = ref + insertSeq;
    else
      altSuffix = insertSeq + ref;
    const char         altSep = (bpB.state == SVBreakendState::RIGHT_OPEN) ? ']' : '[';
    std::ostringstream alt;
    alt << altPrefix << altSep << mateChrom << ':' << matePos << altSep << altSuffix;

    InfoTag_t infoTags;
    infoTags.push_back("SVTYPE=BND");
    infoTags.push_back("MATEID=" + mateId);
    if (isImprecise)
      infoTags.push_back("IMPRECISE");
    else if (_isOutputContig)
      infoTags.push_back("CONTIG=" + sv.contigSeq);
    if (bpARange.size() > 1) infoTags.push_back("CIPOS=" + num((bpARange.begin_pos() + 1) - pos) + "," + num(bpARange.end_pos() - pos));
    if (!isImprecise) addHomologyInfo(chrom, bpARange, 0, infoTags);
    if (!insertSeq.empty()) {
      infoTags.push_back("SVINSLEN=" + num(long(insertSeq.size())));
      infoTags.push_back("SVINSSEQ=" + insertSeq);
    }
    // VcfWriterCandidateSV::modifyTranslocInfo (VcfWriterCandidateSV.cpp:34-45)
    infoTags.push_back("BND_PAIR_COUNT=" + num(bpA.getLocalPairCount()));
    infoTags.push_back("PAIR_COUNT=" + num(bpA.getPairCount()));
    emit(chrom, pos, localId, ref, alt.str(), infoTags);
  }

  /// the single record of an insertion / deletion / tandem duplication (:425-631)
  void writeIndel(const SVCandidate& sv, const SVId& svId, const bool isIndel) const
  {
    const bool              isImprecise(sv.isImprecise());
    const bool              isBp1First = sv.bp1.interval.range.begin_pos() <= sv.bp2.interval.range.begin_pos();
    const SVBreakend&       bpA(isBp1First ? sv.bp1 : sv.bp2);
    const SVBreakend&       bpB(isBp1First ? sv.bp2 : sv.bp1);
    const std::string&      chrom(_header.chrom_data.at(size_t(sv.bp1.interval.tid)).label);
    const known_pos_range2& bpARange(bpA.interval.range);
    const known_pos_range2& bpBRange(bpB.interval.range);
    static const unsigned   maxNonSymbolicRecordSize(1000);
    bool                    isSmallVariant = false;
    if (!isImprecise && isIndel && !sv.isUnknownSizeInsertion) {
      const unsigned deleteSize = unsigned(bpBRange.begin_pos() - bpARange.begin_pos());
      const unsigned insertSize = unsigned(sv.insertSeq.size());
      isSmallVariant            = (deleteSize <= maxNonSymbolicRecordSize) && (insertSize <= maxNonSymbolicRecordSize);
    }
    pos_t internal_pos    = bpARange.center_pos();
    pos_t internal_endPos = bpBRange.center_pos();
    if (!isImprecise) {
      internal_pos    = bpARange.begin_pos();
      internal_endPos = sv.isBreakendRangeSameShift() ? bpBRange.begin_pos() : (bpBRange.end_pos() - 1);
    }
    const pos_t bpAPosAdjust = bpA.getLeftSideOfBkptAdjustment();
    const pos_t pos          = internal_pos + 1 + bpAPosAdjust;
    const pos_t endPos       = internal_endPos + 1 + bpB.getLeftSideOfBkptAdjustment();
    if (pos < 1) return;

    const pos_t       beginRefPos = pos - 1;
    const pos_t       endRefPos   = isSmallVariant ? (endPos - 1) : beginRefPos;
    const std::string ref(refSeq(chrom, beginRefPos, endRefPos));
    if (unsigned(1 + endRefPos - beginRefPos) != ref.size()) {
      std::ostringstream oss;
      oss << "Unexpected reference allele size: " << ref.size() << "\n\tExpected: " << (1 + endRefPos - beginRefPos) << "\n";
      throw GeneralException(oss.str());
    }
    const std::string label(svId.getLabel());
    const std::string alt = isSmallVariant ? (ref.substr(0, 1) + sv.insertSeq) : ("<" + label + ">");

    InfoTag_t infoTags;
    infoTags.push_back("END=" + num(endPos));
    infoTags.push_back("SVTYPE=" + label.substr(0, label.find(':')));
    if (!sv.isUnknownSizeInsertion) {
      const pos_t refLen = endPos - pos;
      pos_t       svLen  = refLen;
      if (isIndel) {
        const pos_t insertLen = pos_t(sv.insertSeq.size());
        svLen                 = (insertLen > refLen) ? insertLen : -refLen;
      }
      infoTags.push_back("SVLEN=" + num(svLen));
    }
    if (isSmallVariant && !sv.insertAlignment.empty()) infoTags.push_back("CIGAR=1M" + ALIGNPATH::apath_to_cigar(sv.insertAlignment));
    if (isImprecise)
      infoTags.push_back("IMPRECISE");
    else if (_isOutputContig)
      infoTags.push_back("CONTIG=" + sv.contigSeq);
    if (bpARange.size() > 1)
      infoTags.push_back("CIPOS=" + num(bpARange.begin_pos() - internal_pos) + "," + num((bpARange.end_pos() - 1) - internal_pos));
    if (!isSmallVariant && bpBRange.size() > 1)
      infoTags.push_back("CIEND=" + num(bpBRange.begin_pos() - internal_endPos) + "," + num((bpBRange.end_pos() - 1) - internal_endPos));
    if (!isImprecise) addHomologyInfo(chrom, bpARange, bpAPosAdjust, infoTags);
    if (!isSmallVariant && !(sv.insertSeq.empty() || sv.isUnknownSizeInsertion)) {
      infoTags.push_back("SVINSLEN=" + num(long(sv.insertSeq.size())));
      std::string ins(sv.insertSeq);
      if (!(isBp1First || (bpA.state != bpB.state))) reverseCompStr(ins);
      infoTags.push_back("SVINSSEQ=" + ins);
    }
    if (sv.isUnknownSizeInsertion) {
      if (!sv.unknownSizeInsertionLeftSeq.empty()) infoTags.push_back("LEFT_SVINSSEQ=" + sv.unknownSizeInsertionLeftSeq);
      if (!sv.unknownSizeInsertionRightSeq.empty()) infoTags.push_back("RIGHT_SVINSSEQ=" + sv.unknownSizeInsertionRightSeq);
    }
    // VcfWriterCandidateSV::modifyInvdelInfo (VcfWriterCandidateSV.cpp:47-56)
    infoTags.push_back("UPSTREAM_PAIR_COUNT=" + num(bpA.getLocalPairCount()));
    infoTags.push_back("DOWNSTREAM_PAIR_COUNT=" + num(bpB.getLocalPairCount()));
    infoTags.push_back("PAIR_COUNT=" + num(bpA.getPairCount()));
    emit(chrom, pos, svId.localId, ref, alt, infoTags);
  }

  RefinerInputSource&    _source;
  const bam_header_info& _header;
  std::ostream&          _os;
  const bool             _isOutputContig;
};

}  // namespace manta_amd
