/* Build-time host tool (plain C, gcc): aie_config bytes on stdin -> normalised aie_params bytes on stdout.
 * ai-economist_amd/_specs.py feeds it the configurations that get a compile-time instance of the step kernel and
 * writes the images into aie_spec_generated.h. */
#include <stdio.h>
#include <stdlib.h>

#include "aie_layout.h"

int main(void) {
  aie_config c;
  static aie_params p;
  char err[256] = "";
  if (fread(&c, sizeof(c), 1, stdin) != 1) { fprintf(stderr, "specgen: short config (%zu bytes expected)\n", sizeof(c)); return 2; }
  if (aie_build_params(&c, &p, NULL, err, sizeof(err)) != AIE_OK) { fprintf(stderr, "specgen: %s\n", err); return 3; }
  aie_spec_normalize(&p);
  if (fwrite(&p, sizeof(p), 1, stdout) != 1) return 4;
  return 0;
}
