// orr_store.h -- internal to libomnirecall_host: the store mirror's data, shared by the service
// (orr_service.cpp) and the Cosmos-JSON importer/exporter (orr_import.cpp).
#pragma once
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace orrh_detail {

struct Chunk {                      // CosmosChunkRecord (CosmosIngestionRecords.cs:19-30)
    std::string id, document_id, content;
    int32_t chunk_index = 0;
    std::vector<float> embedding;   // empty = null
    int64_t created_ticks = 0;
};
struct Document {                   // CosmosDocumentRecord, the fields the path reads
    std::string id, file_name;
    int64_t created_ticks = 0;
};

int fail(int code, const std::string &msg);      // sets the text orrh_last_error() returns on this thread
std::string iso_utc(int64_t ticks);              // DateTime (Kind=Utc) as System.Text.Json writes it
void json_string(const std::string &s, std::string &out);

}  // namespace orrh_detail

struct orrh_store {
    std::mutex mu;
    std::vector<std::string> doc_order;                 // enumeration order of _chunksByDocument
    std::map<std::string, orrh_detail::Document> documents;
    std::map<std::string, std::vector<orrh_detail::Chunk>> chunks_by_document;
    std::map<std::string, uint64_t> chunk_stamp;        // document -> version of its current chunk list
    uint64_t version = 0;                               // any change
    uint64_t chunks_version = 0;                        // changes the index has to follow
};
