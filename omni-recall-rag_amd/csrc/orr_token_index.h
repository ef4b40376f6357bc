// orr_token_index.h -- host-side build of the shard's token -> rows index (see the .cpp).
#pragma once

#include <cstdint>
#include <vector>

namespace orr {

struct TokenIndexHost {
    std::vector<uint8_t> vpool;        // distinct tokens, 16-byte aligned, space padded
    std::vector<uint64_t> vstart;      // [V]
    std::vector<uint32_t> vlen;        // [V]
    std::vector<uint64_t> post_off;    // [V+1]
    std::vector<uint32_t> post_rows;   // [post_off[V]] ascending candidate positions per token
};

// pool/cstart/clen: the lowercased content rows in candidate order (host memory).
void build_token_index(const uint8_t *pool, const uint64_t *cstart, const uint32_t *clen, int64_t n_rows,
                       int n_threads, TokenIndexHost &out);

}  // namespace orr
