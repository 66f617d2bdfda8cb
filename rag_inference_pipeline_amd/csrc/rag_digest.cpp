// rag_digest.cpp — rag_source_digest(): which sources this library was compiled from (include/rag_amd.h).
// A translation unit of its own so that an edit elsewhere recompiles only this file and what was edited.
#include "../../include/rag_amd.h"

#ifndef RAG_AMD_SOURCE_DIGEST
#define RAG_AMD_SOURCE_DIGEST "unknown"
#endif

extern "C" const char* rag_source_digest(void) { return RAG_AMD_SOURCE_DIGEST; }
