/*
 * ref_dump_wrapper.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Translation unit that pulls the reference's header-only payload extractor
 * (/root/reference/packet_dumping.h:87-188, dump_UDP_packet / dump_TCP_packet) into
 * oracle/_ref/libkmpref.so.  The header is found through -I/root/reference (oracle/Makefile);
 * nothing of it is copied here.  The includes below are the ones the reference programs place
 * in front of it (serial.c:6-13) minus <pcap.h>, which the extractor does not use.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <netinet/ip.h>
#include <netinet/if_ether.h>
#include "packet_dumping.h"
