// rm_efndata.h -- efn() energy table loader (efn.c:157-918).
#pragma once
#include <string>
#include <vector>
#include "rm_host.h"

namespace rma {

// RM_getefndata(): read the eleven .dat files of dir into *ed.
bool	load_efndata( const std::string &dir, rma_efndata_t *ed, std::string &err );

// RM_getefn2data(), efn2.c:130: the sixteen .dat files efn2() reads.
bool	load_efn2data( const std::string &dir, rma_efn2data_t *ed, std::string &err );

// efn()'s tables as the device reads them: int16 image of RME_N16 entries (rm_efn_core.h), padded to a multiple of 8;
// the 100 tetraloop keys
void	efn_tables16( const rma_efndata_t *ed, std::vector<int16_t> &t16, std::vector<int32_t> &tlkey );

// efn_datadir parameter, else $EFNDATA (score.c:1584-1590).
std::string	find_efndata_dir( Descriptor &d );

}	// namespace rma
