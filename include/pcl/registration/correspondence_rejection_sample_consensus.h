// pcl/registration/correspondence_rejection_sample_consensus.h - see correspondence_rejection.h
#include <pcl/registration/correspondence_rejection.h>
