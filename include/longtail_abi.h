/* longtail_abi.h -- the slice of longtail's plugin ABI that the HIP plugins implement.
 *
 * When building INSIDE a longtail checkout define LONGTAIL_HIP_USE_LONGTAIL_H and put longtail's
 * `src/` on the include path: the real header is used and this file adds nothing.  Stand-alone (this
 * repository, the GPU box) the declarations below restate -- layout-for-layout, because this is a
 * binary interface -- the three struct-of-function-pointer types and their typedefs:
 *
 *   struct Longtail_API             src/longtail.h:43-46   (first member of every API struct)
 *   struct Longtail_HashAPI         src/longtail.h:203-217
 *   struct Longtail_CompressionAPI  src/longtail.h:262-272
 *   struct Longtail_ChunkerAPI      src/longtail.h:567-594 (feeder :571, ChunkRange :573-578)
 *
 * Every function returns errno-style ints (0 = success); nothing throws or aborts.
 */
#ifndef LONGTAIL_ABI_H
#define LONGTAIL_ABI_H

#if defined(LONGTAIL_HIP_USE_LONGTAIL_H)
#include "longtail.h"
#else

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct Longtail_API;
typedef void (*Longtail_DisposeFunc)(struct Longtail_API* api);
struct Longtail_API
{
    Longtail_DisposeFunc Dispose;
};

/* ---- hash ---- */
struct Longtail_HashAPI;
typedef struct Longtail_HashAPI_Context* Longtail_HashAPI_HContext;
typedef uint32_t (*Longtail_Hash_GetIdentifierFunc)(struct Longtail_HashAPI* hash_api);
typedef int (*Longtail_Hash_BeginContextFunc)(struct Longtail_HashAPI* hash_api, Longtail_HashAPI_HContext* out_context);
typedef void (*Longtail_Hash_HashFunc)(struct Longtail_HashAPI* hash_api, Longtail_HashAPI_HContext context, uint32_t length, const void* data);
typedef uint64_t (*Longtail_Hash_EndContextFunc)(struct Longtail_HashAPI* hash_api, Longtail_HashAPI_HContext context);
typedef int (*Longtail_Hash_HashBufferFunc)(struct Longtail_HashAPI* hash_api, uint32_t length, const void* data, uint64_t* out_hash);
struct Longtail_HashAPI
{
    struct Longtail_API m_API;
    Longtail_Hash_GetIdentifierFunc GetIdentifier;
    Longtail_Hash_BeginContextFunc BeginContext;
    Longtail_Hash_HashFunc Hash;
    Longtail_Hash_EndContextFunc EndContext;
    Longtail_Hash_HashBufferFunc HashBuffer;
};

/* ---- compression ---- */
struct Longtail_CompressionAPI;
typedef size_t (*Longtail_CompressionAPI_GetMaxCompressedSizeFunc)(struct Longtail_CompressionAPI* compression_api, uint32_t settings_id, size_t size);
typedef int (*Longtail_CompressionAPI_CompressFunc)(struct Longtail_CompressionAPI* compression_api, uint32_t settings_id, const char* uncompressed, char* compressed, size_t uncompressed_size, size_t max_compressed_size, size_t* out_compressed_size);
typedef int (*Longtail_CompressionAPI_DecompressFunc)(struct Longtail_CompressionAPI* compression_api, const char* compressed, char* uncompressed, size_t compressed_size, size_t max_uncompressed_size, size_t* out_uncompressed_size);
struct Longtail_CompressionAPI
{
    struct Longtail_API m_API;
    Longtail_CompressionAPI_GetMaxCompressedSizeFunc GetMaxCompressedSize;
    Longtail_CompressionAPI_CompressFunc Compress;
    Longtail_CompressionAPI_DecompressFunc Decompress;
};

/* ---- chunker ---- */
struct Longtail_ChunkerAPI;
typedef struct Longtail_ChunkerAPI_Chunker* Longtail_ChunkerAPI_HChunker;
typedef int (*Longtail_Chunker_Feeder)(void* context, Longtail_ChunkerAPI_HChunker chunker, uint32_t requested_size, char* buffer, uint32_t* out_size);
struct Longtail_Chunker_ChunkRange
{
    const uint8_t* buf;
    uint64_t offset;
    uint32_t len;
};
typedef int (*Longtail_Chunker_GetMinChunkSizeFunc)(struct Longtail_ChunkerAPI* chunker_api, uint32_t* out_min_chunk_size);
typedef int (*Longtail_Chunker_CreateChunkerFunc)(struct Longtail_ChunkerAPI* chunker_api, uint32_t min_chunk_size, uint32_t avg_chunk_size, uint32_t max_chunk_size, Longtail_ChunkerAPI_HChunker* out_chunker);
typedef int (*Longtail_Chunker_NextChunkFunc)(struct Longtail_ChunkerAPI* chunker_api, Longtail_ChunkerAPI_HChunker chunker, Longtail_Chunker_Feeder feeder, void* feeder_context, struct Longtail_Chunker_ChunkRange* out_chunk_range);
typedef int (*Longtail_Chunker_DisposeChunkerFunc)(struct Longtail_ChunkerAPI* chunker_api, Longtail_ChunkerAPI_HChunker chunker);
typedef int (*Longtail_Chunker_NextChunkFromBufferFunc)(struct Longtail_ChunkerAPI* chunker_api, Longtail_ChunkerAPI_HChunker chunker, const void* buffer, uint64_t buffer_size, const void** out_next_chunk_start);
struct Longtail_ChunkerAPI
{
    struct Longtail_API m_API;
    Longtail_Chunker_GetMinChunkSizeFunc GetMinChunkSize;
    Longtail_Chunker_CreateChunkerFunc CreateChunker;
    Longtail_Chunker_NextChunkFunc NextChunk;
    Longtail_Chunker_DisposeChunkerFunc DisposeChunker;
    Longtail_Chunker_NextChunkFromBufferFunc NextChunkFromBuffer;
};

#ifdef __cplusplus
}
#endif
#endif /* LONGTAIL_HIP_USE_LONGTAIL_H */
#endif /* LONGTAIL_ABI_H */
