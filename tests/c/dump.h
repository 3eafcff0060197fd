/* dump.h -- reader of the fixture dumps the Python tests write: <dir>/<name>.bin raw little-endian values
 * (double / int32 / int64), sizes known to the caller through <dir>/meta.txt lines "name count". */
#ifndef NAGP_TEST_DUMP_H
#define NAGP_TEST_DUMP_H
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static size_t dump_count(const char* dir, const char* name) {
  char path[1024], key[128]; size_t cnt; FILE* f;
  snprintf(path, sizeof path, "%s/meta.txt", dir);
  f = fopen(path, "r");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(3); }
  while (fscanf(f, "%127s %zu", key, &cnt) == 2)
    if (!strcmp(key, name)) { fclose(f); return cnt; }
  fclose(f);
  fprintf(stderr, "no entry %s in %s\n", name, path);
  exit(3);
}
static void* dump_load(const char* dir, const char* name, size_t elem, size_t* count) {
  char path[1024]; FILE* f; void* p; size_t n = dump_count(dir, name);
  snprintf(path, sizeof path, "%s/%s.bin", dir, name);
  f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(3); }
  p = malloc((n * elem) > 0 ? n * elem : 1);
  if (fread(p, elem, n, f) != n) { fprintf(stderr, "short read %s\n", path); exit(3); }
  fclose(f);
  if (count) *count = n;
  return p;
}
static double dump_scalar(const char* dir, const char* name) {
  double* p = (double*)dump_load(dir, name, 8, NULL); double v = p[0]; free(p); return v;
}
/* max |a-b| / max |b| over entries finite in b; NaN patterns must agree */
static double rel_diff(const double* a, const double* b, size_t n, const char* what) {
  double mx = 0.0, sc = 0.0; size_t i;
  for (i = 0; i < n; ++i) {
    if (isnan(b[i]) != isnan(a[i])) { fprintf(stderr, "%s: NaN pattern differs at %zu\n", what, i); return INFINITY; }
    if (isnan(b[i])) continue;
    if (isinf(b[i])) { if (a[i] != b[i]) return INFINITY; continue; }
    if (fabs(b[i]) > sc) sc = fabs(b[i]);
    if (fabs(a[i] - b[i]) > mx) mx = fabs(a[i] - b[i]);
  }
  printf("  %-8s max|diff| %.3e  scale %.3e  rel %.3e\n", what, mx, sc, mx / (sc + 1e-300));
  return mx / (sc + 1e-300);
}
#endif
