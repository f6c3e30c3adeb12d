/* main.c -- `lr2rmats <command> [options]` (src/main.c:19-49). */
#include <stdio.h>
#include "l2r_host.h"

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "\nProgram: lr2rmats (Long read to rMATS) -- MI355X build\nUsage:   lr2rmats <command> [options]\n\nCommands: \n");
        fprintf(stderr, "         filter       filter out alignment records with low confidence\n");
        fprintf(stderr, "         update-gtf   generate new GTF file based on BAM/SAM and existing GTF file\n");
        fprintf(stderr, "         unique-gtf   generate GTF file that only contain unique transcript based on BAM/SAM or GTF file\n");
        fprintf(stderr, "         bam2gtf      generate transcript and exon information based on BAM/SAM file\n\n");
        return 1;
    }
    return h_main(argc - 1, argv + 1);
}
