/* oracle_main.c -- TEST INFRASTRUCTURE ONLY (see oracle.h): CLI wrapper. */
#include <stdio.h>
#include "oracle.h"

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "Usage:   lr2rmats_oracle <update-gtf|bam2gtf|unique-gtf> [options]\n"); return 1; }
    return orc_main(argc - 1, argv + 1);
}
