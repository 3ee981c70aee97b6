/* stb_ref.c — TEST INFRASTRUCTURE ONLY.  Compiles the stb codecs the
 * reference vendors (dependencies/stb/stb/stb_image.h v2.27,
 * stb_image_write.h v1.16; used at texture.cpp:34-36,101 and film.cpp:63-78)
 * from where they lie under /root/reference, into oracle/_ref/libstbref.so.
 * Nothing is copied: this file only includes them.  tests/test_image_io.py uses
 * the result to pin the product's own HDR / PNG codecs (hobbyraytracer_amd/host/image_io.cpp). */
#define STB_IMAGE_IMPLEMENTATION
#include <stb/stb_image.h>
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include <stb/stb_image_write.h>
