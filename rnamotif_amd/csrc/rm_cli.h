// rm_cli.h -- command line front end shared by the rnamotif executable:
// everything main() does in /root/reference/src/rnamot.c:42-191 except the
// scan itself, which is delegated to a ScanBackend.
#pragma once
#include "rm_driver.h"

namespace rma {

// Build the scanner for a compiled program; efn is null when the score
// program has no efn() call.  Throws Error on failure.
typedef ScanBackend ( *BackendFactory )( const rma_program_t *prog, const rma_efndata_t *efn, const rma_efn2data_t *efn2 );

// Prepared search: compiled descriptor + flattened program (+ energy tables).
struct Prepared {
	std::unique_ptr<Descriptor>	descr;
	std::unique_ptr<rma_program_t>	prog;
	std::unique_ptr<rma_efndata_t>	efn;	// null if unused
	std::unique_ptr<rma_efn2data_t>	efn2;	// null unless the score section calls efn2()
};

// RM_init .. RM_linkscore, rnamot.c:49-98, plus flattening.
Prepared	prepare( const Args &args );

int	cli_main( int argc, char **argv, BackendFactory make_backend );

}	// namespace rma
