// TEMPORARY: entry points not implemented yet return an error (never a silent fallback).
#include "common.h"
#define NOT_YET(name) return oai::set_error(OAI_ERR_ARG, name ": not implemented yet")
extern "C" {
int oai_unet_create(const oai_layer_params*, float, oai_unet**) { NOT_YET("oai_unet_create"); }
void oai_unet_destroy(oai_unet*) {}
size_t oai_unet_workspace_bytes(const oai_unet*, int, int, int, int) { return 0; }
int oai_unet_forward_tiles(oai_unet*, const float*, float*, int, int, int, int, void*, size_t, void*) { NOT_YET("oai_unet_forward_tiles"); }
int oai_segment_tiles(oai_unet*, const float*, int, int, int, const int*, const int*, int, int, int, float*, int, void*, size_t, void*) { NOT_YET("oai_segment_tiles"); }
int oai_stitch_blocks(const float*, int, int, int, int, const int*, const int*, const int*, float*, void*) { NOT_YET("oai_stitch_blocks"); }
double oai_unet_tile_flops(const oai_unet*, int, int, int, const int*, int) { return 0; }
int oai_icon_create(const oai_icon_unet_params*, int, int, int, oai_icon**) { NOT_YET("oai_icon_create"); }
void oai_icon_destroy(oai_icon*) {}
size_t oai_icon_workspace_bytes(const oai_icon*) { return 0; }
int oai_icon_forward(oai_icon*, const float*, const float*, float*, void*, size_t, void*) { NOT_YET("oai_icon_forward"); }
int oai_icon_unet_forward(oai_icon*, int, const float*, const float*, int, int, int, float*, void*, size_t, void*) { NOT_YET("oai_icon_unet_forward"); }
}
