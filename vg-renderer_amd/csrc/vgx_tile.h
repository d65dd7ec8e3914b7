// vgx_tile.h -- the tile kernel of ordinary batches (vgx_tile.hip): what the host and the other emit kernels need to know of it.
#ifndef VGX_TILE_H
#define VGX_TILE_H

#include "vgx_internal.h"

#define VGX_TILE_ELEMS 2048u /* elements per tile = per 512-thread workgroup (the shape of k_tmpl_emit, vgx_internal.h) */

struct VgxTileRec // 16 bytes
{
	uint32_t mesh0;     // mesh that owns the tile's first element; bit 31: that element is the mesh's element 0
	uint32_t mesh_last; // last mesh with an element in the tile
	uint32_t nel;       // elements in the tile (the batch's last tile is short)
	uint32_t pad;
};

// The batch is emitted by k_emit_tiles (else by k_fill + k_stroke): decided on the device from what the scan over the meshes found.
// k_fill asks the same question to stay out of the way.
__device__ __forceinline__ bool vgx_tile_mode_on(const VgxTotals* T)
{
	return T->status == VGX_OK && T->has_general_stroke == 0u && T->sizes.num_meshes != 0 && T->sizes.num_meshes < 0x7FFFFFFFull;
}

void vgx_launch_emit_tiles(const VgxStrokeArgs& a, VgxTileRec* tiles, uint64_t capTiles, hipStream_t s);

#endif
