/* Generator of tests/golden/png_gray_*.npy (run in the build container only, see gen_png_gray.sh): decodes a PNG to
 * 8-bit grey exactly the way OpenCV's PNG reader does for cv2.imread(path, cv2.IMREAD_GRAYSCALE) (the reference's
 * match.py:118-119) - libpng with png_set_strip_16 / png_set_strip_alpha / png_set_palette_to_rgb and
 * png_set_rgb_to_gray(png, 1, 0.299, 0.587) - and writes the raw bytes.  Linked against the real libpng 1.6.37 of this
 * image, so the fixture pins util.read_gray's restated arithmetic to libpng itself.
 *   gcc gen_png_gray.c -I/opt/conda/include -L/opt/conda/lib -lpng16 -lz -o gen_png_gray */
#include <png.h>
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    png_structp png = png_create_read_struct(PNG_LIBPNG_VER_STRING, NULL, NULL, NULL);
    png_infop info = png_create_info_struct(png);
    if (setjmp(png_jmpbuf(png))) return 4;
    png_init_io(png, f);
    png_read_info(png, info);
    png_uint_32 w, h;
    int depth, ctype;
    png_get_IHDR(png, info, &w, &h, &depth, &ctype, NULL, NULL, NULL);
    if (depth == 16) png_set_strip_16(png);
    if (ctype & PNG_COLOR_MASK_ALPHA) png_set_strip_alpha(png);
    if (ctype == PNG_COLOR_TYPE_PALETTE) png_set_palette_to_rgb(png);
    if ((ctype & PNG_COLOR_MASK_COLOR) == 0 && depth < 8) png_set_expand_gray_1_2_4_to_8(png);
    if (ctype & PNG_COLOR_MASK_COLOR) png_set_rgb_to_gray(png, 1, 0.299, 0.587);
    png_set_interlace_handling(png);
    png_read_update_info(png, info);
    png_bytep *rows = malloc(sizeof(png_bytep) * h);
    unsigned char *buf = malloc((size_t)w * h);
    for (png_uint_32 y = 0; y < h; ++y) rows[y] = buf + (size_t)y * w;
    png_read_image(png, rows);
    png_read_end(png, NULL);
    FILE *o = fopen(argv[2], "wb");
    fwrite(buf, 1, (size_t)w * h, o);
    fclose(o);
    printf("%u %u libpng %s\n", (unsigned)w, (unsigned)h, png_get_libpng_ver(NULL));
    return 0;
}
