// graph_vec.hpp -- host graph types of ./pagerank: GraphVec and SlidingGraphVec.
//
// Same role and public names as the reference's GraphVec.h / SlidingGraphVec.h
// (vertex_count, edge_count, sliding_window_size, edge_batch, new_stream, StreamUpdates,
// SerializeEdgeStream, ConstructGraph), re-designed for a device-resident engine:
//   * the .bin stream (int32 V, then int32 pairs; GraphVec.h:43-70) is mmap'ed once instead
//     of being fread() one integer at a time, so StreamUpdates is two memcpy's;
//   * the window is a position in that mapping; no per-vertex std::vector adjacency is kept
//     (the reference erases from vector fronts, O(degree) per delete) -- the device builds
//     its own CSR, and the host materialises a flat out-CSR only when --validate asks for
//     the power-iteration check.
#pragma once

#include <algorithm>
#include <cassert>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "edge_batch.hpp"
#include "meta.hpp"

class GraphVec {
public:
    GraphVec() = default;
    GraphVec(const std::string &fname, bool di) : filename(fname), directed(di) { MapFile(); }
    virtual ~GraphVec() {
        if (map_) munmap(map_, map_bytes_);
    }
    GraphVec(const GraphVec &) = delete;
    GraphVec &operator=(const GraphVec &) = delete;

    // stream edge k = (src(k), dst(k))
    IndexType src(size_t k) const { return pairs_[2 * k]; }
    IndexType dst(size_t k) const { return pairs_[2 * k + 1]; }
    size_t stream_length() const { return stream_len_; }

    virtual bool StreamUpdates(size_t) { return true; }

    // flat out-CSR of stream edges [lo, hi) (mirrored when undirected); used by validation only
    void BuildOutCSR(size_t lo, size_t hi, std::vector<IndexType> &row_ptr, std::vector<IndexType> &col) const {
        row_ptr.assign((size_t)vertex_count + 1, 0);
        for (size_t k = lo; k < hi; ++k) {
            row_ptr[(size_t)src(k) + 1]++;
            if (!directed) row_ptr[(size_t)dst(k) + 1]++;
        }
        for (IndexType v = 0; v < vertex_count; ++v) row_ptr[(size_t)v + 1] += row_ptr[v];
        col.resize((size_t)row_ptr[vertex_count]);
        std::vector<IndexType> fill(row_ptr.begin(), row_ptr.end() - 1);
        for (size_t k = lo; k < hi; ++k) { // stream order, like GraphVec.h:60-68
            col[(size_t)fill[src(k)]++] = dst(k);
            if (!directed) col[(size_t)fill[dst(k)]++] = src(k);
        }
    }

    std::string filename;
    bool directed = true;
    IndexType vertex_count = 0;
    IndexType edge_count = 0; // directed edges of the (window) graph

protected:
    void MapFile() {
        std::cout << "read filename=" << filename << std::endl;
        int fd = open(filename.c_str(), O_RDONLY);
        if (fd < 0) {
            std::cout << "cannot open " << filename << std::endl;
            std::exit(-1);
        }
        struct stat st;
        fstat(fd, &st);
        map_bytes_ = (size_t)st.st_size;
        if (map_bytes_ < sizeof(IndexType)) {
            std::cout << "file too short: " << filename << std::endl;
            std::exit(-1);
        }
        map_ = mmap(nullptr, map_bytes_, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (map_ == MAP_FAILED) {
            std::cout << "mmap failed: " << filename << std::endl;
            std::exit(-1);
        }
        const IndexType *words = static_cast<const IndexType *>(map_);
        vertex_count = words[0];
        pairs_ = words + 1;
        stream_len_ = (map_bytes_ - sizeof(IndexType)) / sizeof(IndexType) / 2;
        std::cout << "vertex_count=" << vertex_count << std::endl;
    }

    void *map_ = nullptr;
    size_t map_bytes_ = 0;
    const IndexType *pairs_ = nullptr;
    size_t stream_len_ = 0;
};

class SlidingGraphVec : public GraphVec {
public:
    SlidingGraphVec(const std::string &fname, bool di) : GraphVec(fname, di) {
        PrepareSlidingGraph();
        edge_batch = new EdgeBatch((IndexType)(gStreamUpdateCountPerBatch * 4));
        new_stream = new EdgeBatch((IndexType)(gStreamUpdateCountPerBatch * 2));
    }
    ~SlidingGraphVec() override {
        delete edge_batch;
        delete new_stream;
    }

    // Derive window / batch sizes from the flags: the arithmetic (and its truncating
    // conversions) of SlidingGraphVec.h:46-66, and the same log lines.
    void PrepareSlidingGraph() {
        const size_t total = stream_len_;
        sliding_window_size = (IndexType)((double)total * gWindowRatio);
        if (gWorkloadConfigType == SLIDE_WINDOW_RATIO) {
            gStreamUpdateCountPerBatch = (size_t)(gStreamUpdateCountVersusWindowRatio * sliding_window_size);
            gStreamUpdateCountTotal = gStreamUpdateCountPerBatch * gStreamBatchCount;
        } else {
            gStreamBatchCount = (gStreamUpdateCountTotal + gStreamUpdateCountPerBatch - 1) / gStreamUpdateCountPerBatch;
        }
        if (gStreamUpdateCountTotal > total - (size_t)sliding_window_size)
            gStreamUpdateCountTotal = total - (size_t)sliding_window_size;
        std::cout << "after workload config: gStreamUpdateCountPerBatch=" << gStreamUpdateCountPerBatch
                  << ",gStreamBatchCount=" << gStreamBatchCount << ",gStreamUpdateCountTotal=" << gStreamUpdateCountTotal
                  << std::endl;
        edge_count = directed ? sliding_window_size : sliding_window_size * 2;
        std::cout << "sliding window size=" << sliding_window_size
                  << ",gStreamUpdateCountPerBatch=" << gStreamUpdateCountPerBatch << std::endl;
        std::cout << "edge_count=" << edge_count << std::endl;
        for (size_t k = 0; k < (size_t)sliding_window_size; ++k) {
            if (src(k) < 0 || src(k) >= vertex_count || dst(k) < 0 || dst(k) >= vertex_count) {
                std::cout << "vertex id out of range at stream edge " << k << std::endl;
                std::exit(-1);
            }
        }
        pos = (size_t)sliding_window_size;
    }

    // Current window in stream order, not mirrored (SlidingGraphVec.h:201-217).
    void SerializeEdgeStream(EdgeBatch *out) const {
        assert(out->size >= sliding_window_size);
        const size_t lo = pos - (size_t)sliding_window_size;
        for (size_t k = lo; k < pos; ++k) {
            out->edge1[k - lo] = src(k);
            out->edge2[k - lo] = dst(k);
        }
        out->length = sliding_window_size;
    }

    // Advance the window by stream_count edges and fill new_stream / edge_batch:
    // [c deletes][c inserts] and, for undirected streams, the mirrored copy
    // (SlidingGraphVec.h:219-275). Returns true when fewer than stream_count edges remain; the
    // partial batch is dropped, as in the reference.
    bool StreamUpdates(size_t stream_count) override {
        if (pos + stream_count > stream_len_) return true;
        const size_t c = stream_count, lo = pos - (size_t)sliding_window_size;
        for (size_t i = 0; i < c; ++i) {
            const IndexType a = src(pos + i), b = dst(pos + i);
            if (a < 0 || a >= vertex_count || b < 0 || b >= vertex_count) {
                std::cout << "vertex id out of range at stream edge " << pos + i << std::endl;
                std::exit(-1);
            }
            new_stream->edge1[i] = a;
            new_stream->edge2[i] = b;
            new_stream->is_insert[i] = 1;
            edge_batch->edge1[i] = src(lo + i);
            edge_batch->edge2[i] = dst(lo + i);
            edge_batch->is_insert[i] = 0;
            edge_batch->edge1[c + i] = a;
            edge_batch->edge2[c + i] = b;
            edge_batch->is_insert[c + i] = 1;
        }
        new_stream->length = (IndexType)c;
        pos += c;
        size_t len = 2 * c;
        if (!directed) {
            std::memcpy(edge_batch->edge1 + len, edge_batch->edge2, sizeof(IndexType) * len);
            std::memcpy(edge_batch->edge2 + len, edge_batch->edge1, sizeof(IndexType) * len);
            std::memcpy(edge_batch->is_insert + len, edge_batch->is_insert, len);
            len *= 2;
        }
        edge_batch->length = (IndexType)len;
        return false;
    }

    // out-CSR of the current window (the reference's ConstructGraph == ScratchConstructWindowGraph)
    void ConstructGraph(std::vector<IndexType> &row_ptr, std::vector<IndexType> &col) const {
        BuildOutCSR(pos - (size_t)sliding_window_size, pos, row_ptr, col);
    }

    IndexType sliding_window_size = 0;
    size_t pos = 0;                 // stream edges consumed; the window is [pos - W, pos)
    EdgeBatch *edge_batch = nullptr; // directed records with insert/delete flags
    EdgeBatch *new_stream = nullptr; // the c new stream edges, undirected-agnostic
};
