// edge_batch.hpp -- SoA batch of directed edge records (insert / delete), the host-side
// container handed to dppr_set_batch. Same public members as the reference's EdgeBatch
// (EdgeBatch.h:6-30): edge1, edge2, is_insert, length, size.
#pragma once

#include <cstdint>
#include <memory>

#include "meta.hpp"

struct EdgeBatch {
    explicit EdgeBatch(IndexType capacity)
        : size(capacity), length(0), e1_(new IndexType[capacity > 0 ? capacity : 1]),
          e2_(new IndexType[capacity > 0 ? capacity : 1]), ins_(new uint8_t[capacity > 0 ? capacity : 1]),
          edge1(e1_.get()), edge2(e2_.get()), is_insert(ins_.get()) {}
    EdgeBatch(const EdgeBatch &) = delete;
    EdgeBatch &operator=(const EdgeBatch &) = delete;

    IndexType size;   // capacity
    IndexType length; // records in use

private:
    std::unique_ptr<IndexType[]> e1_, e2_;
    std::unique_ptr<uint8_t[]> ins_;

public:
    IndexType *edge1;
    IndexType *edge2;
    uint8_t *is_insert; // 1 = insert, 0 = delete (the reference stores bool; same bytes)
};
