/* ORACLE: temporary stubs for codecs not restated yet (replaced as they land). */
#include "kzo.h"
int kzo_fpaq_encode(kzo_obs* s, const uint8_t* b, int n) { (void)s; (void)b; (void)n; return -1; }
int kzo_fpaq_decode(kzo_ibs* s, uint8_t* b, int n) { (void)s; (void)b; (void)n; return -1; }
int kzo_srt_forward(const uint8_t* s, int n, uint8_t* d, int c, int* p) { (void)s; (void)n; (void)d; (void)c; *p = 0; return 0; }
int kzo_srt_inverse(const uint8_t* s, int n, uint8_t* d, int c, int* p) { (void)s; (void)n; (void)d; (void)c; *p = 0; return 0; }
int kzo_lz_forward(int x, int t, const uint8_t* s, int n, uint8_t* d, int c, int* p) { (void)x; (void)t; (void)s; (void)n; (void)d; (void)c; *p = 0; return 0; }
int kzo_lz_inverse(int x, const uint8_t* s, int n, uint8_t* d, int c, int* p) { (void)x; (void)s; (void)n; (void)d; (void)c; *p = 0; return 0; }
