/* oracle/stubs/xxhash.h -- TEST INFRASTRUCTURE ONLY.  The reference's src/simd/distances_ref.cc includes xxhash.h for one
 * function outside the search path (calculate_hash_ref); the library is not in this image.  This declaration lets that
 * translation unit compile where it lies; ref_simd.cpp defines the symbol as a trap. */
#pragma once
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
uint64_t XXH3_64bits(const void* data, size_t len);
#ifdef __cplusplus
}
#endif
