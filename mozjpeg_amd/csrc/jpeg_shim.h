/* jpeg_shim.h -- what jpeg_shim.c and jpeg_api.c share (both are part of the libjpeg drop-in, not of the public ABI) */
#ifndef MJH_JPEG_SHIM_H
#define MJH_JPEG_SHIM_H
/* forget the compression in flight on this object, if any (staged image, encoder lease); returns 1 if there was one */
int mjh_shim_drop(void *cinfo);
#endif
