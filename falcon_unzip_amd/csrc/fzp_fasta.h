// fzp_fasta.h -- the FASTA records of a group of files, indexed on the device (fzp_fasta.hip)
#pragma once
#include "fzp_common.h"

struct FaIndex {
    int64_t n_rec = 0, n_lines = 0, join_bytes = 0;
    // per record, file order (host): the file it lies in, its bases, where its name (the header's first word) stands in the raw buffer
    std::vector<int32_t> h_file;
    std::vector<int64_t> h_len, h_name_b, h_name_e;
    // the names themselves, back to back (record r at h_names[h_noff[r] .. h_noff[r + 1])): gathered on the device, so that the host needs no copy of the files' bytes
    std::vector<int64_t> h_noff;
    std::vector<char> h_names;
    // per record (device): begin / end of its bases as offsets from the raw buffer's first byte -- into the raw buffer where the record is one line, else into d_join
    DevBuf<int64_t> d_be;
    DevBuf<uint8_t> d_join;
};
// d_raw: the files' bytes on the device, file t at foff[t] (ascending), a '\n' behind every file; n_bytes includes the last one.  Waits for the stream (the host vectors).
int fzp_fasta_index_dev(fzp_ctx *ctx, hipStream_t st, const uint8_t *d_raw, int64_t n_bytes, const int64_t *foff, int nf, FaIndex &X);
// test hook: the sequences of records first .. first + n as the packer will read them, joined back to back into host_out (sum of their h_len bytes)
int fzp_fasta_fetch_seqs(fzp_ctx *ctx, hipStream_t st, const uint8_t *d_raw, const FaIndex &X, int64_t first, int64_t n, uint8_t *host_out);
