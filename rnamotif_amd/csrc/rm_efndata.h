// rm_efndata.h -- efn() energy table loader (efn.c:157-918).
#pragma once
#include <string>
#include "rm_host.h"

namespace rma {

// RM_getefndata(): read the eleven .dat files of dir into *ed.
bool	load_efndata( const std::string &dir, rma_efndata_t *ed, std::string &err );

// RM_getefn2data(), efn2.c:130: the sixteen .dat files efn2() reads.
bool	load_efn2data( const std::string &dir, rma_efn2data_t *ed, std::string &err );

// efn_datadir parameter, else $EFNDATA (score.c:1584-1590).
std::string	find_efndata_dir( Descriptor &d );

}	// namespace rma
