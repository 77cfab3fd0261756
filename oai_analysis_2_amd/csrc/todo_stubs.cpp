// TEMPORARY: entry points not implemented yet return an error (never a silent fallback).
#include "common.h"
#define NOT_YET(name) return oai::set_error(OAI_ERR_ARG, name ": not implemented yet")
extern "C" {
int oai_icon_create(const oai_icon_unet_params*, int, int, int, oai_icon**) { NOT_YET("oai_icon_create"); }
void oai_icon_destroy(oai_icon*) {}
size_t oai_icon_workspace_bytes(const oai_icon*) { return 0; }
int oai_icon_forward(oai_icon*, const float*, const float*, float*, void*, size_t, void*) { NOT_YET("oai_icon_forward"); }
int oai_icon_unet_forward(oai_icon*, int, const float*, const float*, int, int, int, float*, void*, size_t, void*) { NOT_YET("oai_icon_unet_forward"); }
}
