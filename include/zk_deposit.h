/* Drop-in for libsnark-vnt/src/deposit/depositcgo.hpp. */
#ifndef ZK_DEPOSIT_H
#define ZK_DEPOSIT_H
#include "zk_common.h"
#ifdef __cplusplus
extern "C" {
#endif
char *genCMTS(uint64_t value_s, char *pk_string, char *r_s_string, char *sn_old_string);   /* libsnark-vnt/src/deposit/depositcgo.cpp:271-284 */
/* replaces libsnark-vnt/src/deposit/depositcgo.cpp:302-325: root of the depth-8 commitment tree after appending n leaves given as n concatenated 66-character items */
char *genRoot(char *cmtarray, int n);
/* replaces libsnark-vnt/src/deposit/depositcgo.cpp:327-444; key file /usr/local/prfKey/depositpk.txt.  RT is ignored (the root is recomputed, :402-403) */
char *genDepositproof(uint64_t value, uint64_t value_old, char *sn_old_string, char *r_old_string, char *sn_string, char *r_string, char *sns_string, char *rs_string,
                      char *cmtB_old_string, char *cmtB_string, uint64_t value_s, char *pk_string, char *sn_A_oldstring, char *cmtS_string, char *cmtarray, int n, char *RT, char *sk_string);
/* replaces libsnark-vnt/src/deposit/depositcgo.cpp:446-551; key file /usr/local/prfKey/depositvk.txt */
bool verifyDepositproof(char *data, char *RT, char *pk, char *cmtb_old, char *snold, char *cmtb, char *sns);
#ifdef __cplusplus
}
#endif
#endif
