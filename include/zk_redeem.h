/* Drop-in for libsnark-vnt/src/redeem/redeemcgo.hpp. */
#ifndef ZK_REDEEM_H
#define ZK_REDEEM_H
#include "zk_common.h"
#ifdef __cplusplus
extern "C" {
#endif
/* replaces libsnark-vnt/src/redeem/redeemcgo.cpp:269-322; key file /usr/local/prfKey/redeempk.txt */
char *genRedeemproof(uint64_t value, uint64_t value_old, char *sn_old_string, char *r_old_string, char *sn_string, char *r_string,
                     char *cmtA_old_string, char *cmtA_string, uint64_t value_s, char *sk_string);
/* replaces libsnark-vnt/src/redeem/redeemcgo.cpp:324-418; key file /usr/local/prfKey/redeemvk.txt */
bool verifyRedeemproof(char *data, char *cmtA_old_string, char *sn_old_string, char *cmtA_string, uint64_t value_s);
#ifdef __cplusplus
}
#endif
#endif
